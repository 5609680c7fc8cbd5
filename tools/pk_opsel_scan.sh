#!/bin/bash
# Lists the kernels of hpfg_amd/csrc whose gfx950 ISA contains a packed fp32 VALU instruction that takes an operand of the LOW result half from
# the HIGH dword of a register pair (op_sel:[..1..]) -- the operand form that made first_wgrad_kernel irreproducible beside other streams'
# kernels (common.h: HPFG_NO_PK_F32).  Compile only, no GPU.  Expected output after the attribute: nothing but the header line.
cd "$(dirname "$0")/../hpfg_amd/csrc" || exit 1
out=${TMPDIR:-/tmp}/pk_scan_$$; mkdir -p $out
ls *.hip | xargs -P 6 -I{} sh -c "hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only {} -o $out/{}.s 2>/dev/null"
echo "# kernels with v_pk_*_f32 ... op_sel cross-half operands:"
for f in $out/*.s; do
  awk -v F="$(basename $f .s)" '/^_Z[A-Za-z0-9_]*:/ {name=$1} /v_pk_[a-z]*_f32.*op_sel:\[[01,]*1/ {a[name]++} END {for (n in a) print F, a[n], n}' $f
done | while read f c n; do echo "$f $c $(echo $n | tr -d ':' | c++filt | cut -c1-120)"; done
rm -rf $out
