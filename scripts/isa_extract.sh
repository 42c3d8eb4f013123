#!/bin/bash
# Device ISA of kernels.hip -> /tmp/kernels.s, and one function per file /tmp/isa/<mangled>.s
set -e
cd "$(dirname "$0")/../rust-raytracer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math ${DEFS} -S --cuda-device-only kernels.hip -o /tmp/kernels.s
mkdir -p /tmp/isa && rm -f /tmp/isa/*.s
awk '/^_ZN3rtc[A-Za-z0-9_]*:/{name=$1; sub(":","",name); out="/tmp/isa/" name ".s"} /^\.Lfunc_end/{out=""} out!=""{print > out}' /tmp/kernels.s
for f in /tmp/isa/*trace_kernel*; do
  n=$(basename $f .s)
  printf "%-90s lines %5d valu %5d salu %5d vmem %4d lds %3d\n" "$(echo $n | sed "s/_ZN3rtc12trace_kernelIL//; s/EEvNS_8SceneDev.*//")" $(wc -l < $f) $(grep -c "^\sv_" $f) $(grep -c "^\ss_" $f) $(grep -c "^\s\(global\|flat\|buffer\)_" $f) $(grep -c "^\sds_" $f)
done
