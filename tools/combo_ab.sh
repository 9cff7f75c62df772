run() { env "$@" timeout 100 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-breakdown $EXTRA 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f host %.4f' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"; }
for rep in 1 2; do
echo "base:            $(EXTRA= run X=1)"
echo "no-overlap:      $(EXTRA=--no-overlap run X=1)"
echo "single-stream:   $(EXTRA= run AA_TRAIN_SINGLE_STREAM=1)"
echo "both:            $(EXTRA=--no-overlap run AA_TRAIN_SINGLE_STREAM=1)"
done
