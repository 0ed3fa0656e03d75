// EXPERIMENT (round 2, rejected — see DESIGN.md §9 and profiles/r02_fused_phase_experiment.txt): parameters of the fused
// decode-phase kernel.  Not part of libamt_hip.so.
#pragma once
#include "../../video2music_amd/csrc/kernels.h"

// ---------------- fused decode phase (decode_phase.hip) ----------------
// ONE launch = the skinny GEMM of a decode-step phase (16 x 16 output tiles, 512-thread workgroups) PLUS the workgroups of
// the attention that consumes its result: those start at once, pull their K/V rows into registers / LDS while the GEMM
// tiles run, wait on an arrival counter for the tiles of their clip's row block, and then only have the arithmetic
// left.  a.B == 0: GEMM only.  The GEMM part takes the folded chain's launches only: mode 0, two-source rows,
// column split, pro in {0, 1}; the attention part the folded prologues (fold_u != null).
struct DecodePhaseParams {
    DecodeGemmParams g;
    AttnDecodeParams a;
    unsigned* sync;             // [0..1] arrivals per 16-row block, [2] exit ticket of the attention workgroups, [3] error flag;
                                // zero between launches (the last attention workgroup to leave clears [0..2])
    unsigned long long* stamps; // diagnostic builds only (-DAMT_STAMPS): [workgroup][8] s_memrealtime stamps
    // L2 prefetch blocks (pf_n > 0; exclusive with the attention part): block j touches the first pf_rows rows (pf_pos != null:
    // min(pf_rows, *pf_pos + pf_pos_add)) of K and V of (clip, head) j = b*H + h, i.e. of the NEXT launch's attention
    // workgroup with the same linear block id, which round-robin dispatch places on the same XCD: that launch then finds
    // this share of its K/V in its XCD's L2 (measured: 19 MB touched ahead take 2 us off a 39 MB cross-attention launch)
    const float* pf_k; const float* pf_v; size_t pf_head_stride; int pf_n, pf_rows, pf_row_floats, pf_pos_add; const int* pf_pos;
};
int32_t amt_launch_decode_phase(const DecodePhaseParams& p, hipStream_t stream);
// whether the fused phase kernel takes this model shape (else the separate kernels run)
bool amt_decode_phase_supported(int d, int dff, int hd, int scap);

