#!/bin/bash
# build/variants/libvvhip_<name>.so with extra hipcc flags: tools/variant.sh name "-DVV_WPB=4 ..." [all]
# (flags apply to gemv.hip; with a third argument "all" every source is rebuilt with them)
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"; C=$R/vibevoice_amd/csrc; O=$R/build/obj; V=$R/build/variants
mkdir -p $O $V
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=16"
SFX=""; XF=""
if [ "$3" = "all" ]; then SFX="_$1"; XF="$2"; fi
for f in gemm tile gemv16p headtail prefill attn misc block1d engine; do
  if [ ! -f $O/$f$SFX.o ] || [ $C/$f.hip -nt $O/$f$SFX.o ] || [ $C/vv_common.h -nt $O/$f$SFX.o ]; then /opt/rocm/bin/hipcc $FL $XF -c $C/$f.hip -o $O/$f$SFX.o & fi
done
/opt/rocm/bin/hipcc $FL $2 -c $C/gemv.hip -o $O/gemv_$1.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/gemm$SFX.o $O/tile$SFX.o $O/gemv16p$SFX.o $O/headtail$SFX.o $O/prefill$SFX.o $O/attn$SFX.o $O/misc$SFX.o $O/block1d$SFX.o $O/engine$SFX.o $O/gemv_$1.o -o $V/libvvhip_$1.so
echo $V/libvvhip_$1.so
