for m in ${MODES:-0 2 3 4}; do
  echo "=== AA_EXP_MERGED_FWD=$m"
  AA_EXP_MERGED_FWD=$m timeout 200 python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-breakdown --no-inloop --no-other-configs --host-profile 200 2>&1 >/dev/null | grep -E "host time per iteration|GPU timeline"
done
