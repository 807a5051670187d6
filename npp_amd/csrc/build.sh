#!/bin/bash
# Build libnpp_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libnpp_hip.so
SRCS="api.hip conv_igemm.hip conv_s1.hip conv_g8.hip conv_g4.hip conv_h3.hip conv_thin.hip conv_c32.hip conv_wgrad.hip conv_wgrad_s1.hip conv_wgrad_g4.hip conv_wgrad_h3.hip dwconv.hip bn.hip bn_one.hip pool.hip se.hip bilinear.hip misc.hip loss.hip optim.hip eval.hip targets.hip comm.hip p2p.hip"
OBJS=""
mkdir -p build
# the hash of the sources this library is built from (npp_amd._lib.kernel_source_hash) goes into npp_version(): bench.py reports a
# measured HBM traffic only when sources, library and the committed counters all carry the same hash
SRC_HASH=$(python3 - <<'PY'
import hashlib, os
h = hashlib.sha256()
files = [f for f in sorted(os.listdir(".")) if f.endswith((".hip", ".h"))]
for f in files:
    h.update(f.encode()); h.update(open(f, "rb").read())
h.update(b"npp_hip.h"); h.update(open("../../include/npp_hip.h", "rb").read())
print(h.hexdigest()[:16])
PY
)
if [ "$(cat build/src_hash.txt 2>/dev/null)" != "$SRC_HASH" ]; then echo "$SRC_HASH" > build/src_hash.txt; rm -f build/api.o; fi
pids=()
for s in $SRCS; do
  [ -f "$s" ] || continue
  o=build/${s%.hip}.o
  OBJS="$OBJS $o"
  stale=0      # any header newer than the object: rebuild (a struct shared by two translation units must never come from two versions)
  for h in *.h ../../include/npp_hip.h; do [ "$h" -nt "$o" ] && stale=1; done
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ $stale = 1 ]; then
    hipcc --offload-arch=gfx950 -O3 -fPIC -munsafe-fp-atomics -std=c++17 -Wno-unused-result -DNPP_SRC_HASH=\"$SRC_HASH\" $NPP_EXTRA_HIPCC_FLAGS -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS -ldl
echo "built $OUT"
