#!/bin/bash
# Builds agents_amd/_variants/lib_<name>.so = the current tree with ONE source taken from a git
# revision, for same-box A/B runs of two kernel versions (tools/ab_bench.py AA_LIB_PATH a b).
#   tools/build_variant.sh <name> <git-rev> <csrc file> [more csrc files ...]
set -euo pipefail
name=$1; rev=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/agents_amd/_variants; mkdir -p "$out/obj_$name"
python -c "import sys; sys.path.insert(0, '$root'); from agents_amd import _build; _build.build()" > /dev/null
objs=()
for o in "$root"/agents_amd/_build_obj/*.o; do objs+=("$o"); done
for f in "$@"; do
  tmp=$out/obj_$name/$f
  git -C "$root" show "$rev:agents_amd/csrc/$f" > "$tmp"
  extra=$(python -c "import sys; sys.path.insert(0, '$root'); from agents_amd import _build; print(' '.join(dict(_build.SOURCES)['$f']))")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$root/include" -I"$root/agents_amd/csrc" \
    $extra -c -x hip "$tmp" -o "$tmp.o"
  for i in "${!objs[@]}"; do
    [[ "${objs[$i]}" == */"${f%.hip}.o" ]] && objs[$i]="$tmp.o"
  done
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/lib_$name.so" "${objs[@]}"
echo "$out/lib_$name.so"
