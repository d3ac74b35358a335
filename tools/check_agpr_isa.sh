#!/bin/bash
# Audit of the hard-register AGPR kernel: compiles attention.hip to ISA and lists every v_accvgpr_* that hipcc itself emitted
# (outside ;;#ASMSTART .. ;;#ASMEND) inside attn_bwd_dkv3_kernel - there must be none (attn_agpr.inc).  Exit 1 otherwise.
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -I$R/rlaif-v_amd/csrc -S --cuda-device-only -o $T/attn.s $R/rlaif-v_amd/csrc/attention.hip 2>/dev/null || exit 2
awk '/^_ZN12_GLOBAL__N_120attn_bwd_dkv3_kernel.*:$/{k=1} k&&/s_endpgm/{k=0} k&&/#ASMSTART/{a=1} k&&/#ASMEND/{a=0} k&&!a&&/v_accvgpr|scratch_/{print; bad++} END{print "compiler-emitted accvgpr/scratch instructions in dkv3:", bad+0; exit bad>0}' $T/attn.s
