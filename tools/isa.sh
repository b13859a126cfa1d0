# Diagnostic: gfx950 ISA of the small-file kernel (or another .hip of csrc/) and its instruction counts.
#   bash tools/isa.sh [file.hip] [kernel-name-regex] [extra -D flags]     -> /tmp/isa/<file>.s, /tmp/isa/k.s (the chosen kernel)
F=${1:-mzd_lds.hip}; K=${2:-mzd_lds_kernelILi4ELb0}; shift; shift
mkdir -p /tmp/isa
cd "$(dirname "$0")/../fuse_zstd_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S "$@" -o /tmp/isa/$F.s $F 2>/dev/null || { echo "compile failed"; exit 1; }
awk "/^_ZN.*$K.*:/,/s_endpgm/" /tmp/isa/$F.s > /tmp/isa/k.s
echo "kernel lines $(wc -l < /tmp/isa/k.s)  valu $(grep -c '^\s*v_' /tmp/isa/k.s)  salu $(grep -c '^\s*s_' /tmp/isa/k.s)  ds $(grep -c '^\s*ds_' /tmp/isa/k.s)  vmem $(grep -c '^\s*global_\|^\s*flat_\|^\s*buffer_' /tmp/isa/k.s)  add0 $(grep -c 'v_add_u32_e32 v[0-9]*, 0, v' /tmp/isa/k.s)"
grep -E "^; (TotalNumVgprs|Occupancy|ScratchSize)|^_ZN3mzd.*kernel.*:" /tmp/isa/$F.s | paste - - - - 2>/dev/null | sed 's/; @.*E:/ /' | cut -c1-200
