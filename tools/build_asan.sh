#!/bin/bash
# Sanitizer build of the HOST side of the C ABI: model.hip (graph, folding, packing, arena, plan cache) and
# ftc_api.hip (validation, dispatch) compiled with -fsanitize=address,undefined (device code not instrumented), linked with the
# regular kernel objects into findtextcenternet_amd/csrc/build/asan/libftc_hip_asan.so, plus the host-only C client
# tests/c_abi/ftc_c_host_check.c.  Run by tests/test_c_abi.py::test_host_side_of_the_abi_under_asan_ubsan.
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CS="$ROOT/findtextcenternet_amd/csrc"
OUT="$CS/build/asan"
mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g"
for f in model ftc_api; do
  if [ ! -f "$OUT/$f.o" ] || [ "$CS/$f.hip" -nt "$OUT/$f.o" ] || [ "$ROOT/include/ftc.h" -nt "$OUT/$f.o" ]; then
    EXTRA=""; [ "$f" = model ] && EXTRA="-ffp-contract=off"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++20 -fPIC $SAN $EXTRA -c "$CS/$f.hip" -o "$OUT/$f.o"
  fi
done
OBJS=$(ls "$CS"/build/*.o | grep -v -e '/model.o$' -e '/ftc_api.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o "$OUT/libftc_hip_asan.so" $OBJS "$OUT/model.o" "$OUT/ftc_api.o"
/opt/rocm/lib/llvm/bin/clang -std=c11 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I"$ROOT/include" "$ROOT/tests/c_abi/ftc_c_host_check.c" \
  -o "$OUT/ftc_c_host_check" -L"$OUT" -lftc_hip_asan -Wl,-rpath,"$OUT" -Wl,-rpath,/opt/rocm/lib
echo "$OUT/ftc_c_host_check"
