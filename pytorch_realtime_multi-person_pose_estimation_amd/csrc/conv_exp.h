// Developer ablation switches of the two conv kernels (conv_mfma.hip, conv_mfma_bf16.hip), all in
// one place.  PRODUCTION builds (csrc/Makefile) do not define RTPOSE_DEV_BUILD: every macro below
// collapses to the measured-best setting, no environment variable is read, and a stray
// -DRTPOSE_EXP_* is a compile error.  Developer builds (tools/exp_variants*.sh, tools/ab_env*.sh:
// -DRTPOSE_DEV_BUILD [-DRTPOSE_EXP_...]) switch single load streams / pipeline stages off to
// measure what each costs (verdicts are recorded next to the code they concern and in DESIGN.md §8).
#pragma once
#include <cstdlib>

#ifndef RTPOSE_DEV_BUILD
#if defined(RTPOSE_EXP_TIMELINE) || defined(RTPOSE_EXP_NO_A) || defined(RTPOSE_EXP_NO_B) ||             \
    defined(RTPOSE_EXP_NO_STAGE) || defined(RTPOSE_EXP_NO_FILL) || defined(RTPOSE_EXP_NO_STORE) ||      \
    defined(RTPOSE_EXP_SCALAR_STORE) || defined(RTPOSE_EXP_BSPREAD) || defined(RTPOSE_EXP_HALF_B_ON) || \
    defined(RTPOSE_EXP_STAGGER) || defined(RTPOSE_EXP_TB1X1) || defined(RTPOSE_EXP_HD) ||               \
    defined(RTPOSE_EXP_RB2) || defined(RTPOSE_EXP_W7_PF) || defined(RTPOSE_EXP_W7_TMASK) || defined(RTPOSE_EXP_W7_LPS) || defined(RTPOSE_EXP_W_EPI) || \
    defined(RTPOSE_EXP_W3_SWOLD) || defined(RTPOSE_EXP_W7_XSPLIT) || defined(RTPOSE_EXP_W7_PRIO) || defined(RTPOSE_EXP_W7_FS2SETS) || defined(RTPOSE_EXP_W7_CGMAJOR) || defined(RTPOSE_EXP_TIMELINE3) || \
    defined(RTPOSE_EXP_W4_NOXF) || defined(RTPOSE_EXP_W4_NOLOAD) || defined(RTPOSE_EXP_W4_XCDMAP) || defined(RTPOSE_EXP_W4_A3) || defined(RTPOSE_EXP_W4_PRIO) || defined(RTPOSE_EXP_TIMELINE4) || defined(RTPOSE_EXP_W7_NULLDESC) || defined(RTPOSE_EXP_W4_AUX) || defined(RTPOSE_EXP_W4_L0) || defined(RTPOSE_EXP_STAGE_NEAR) || \
    defined(RTPOSE_EXP_TIMELINE_UNIT)
#error "RTPOSE_EXP_* ablation switches need -DRTPOSE_DEV_BUILD (they are not part of production builds)"
#endif
#endif

namespace rtpose {
// Environment knobs (RTPOSE_CONV_*, RTPOSE_BF16_*) exist in developer builds only.
inline const char* dev_env(const char* name) {
#ifdef RTPOSE_DEV_BUILD
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
}  // namespace rtpose

// per-block s_memtime stamps (tools/timeline_*.py)
#ifdef RTPOSE_EXP_TIMELINE
#define RTPOSE_TSTAMP(slot) \
  if (A.dbg && threadIdx.x == 0) A.dbg[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime()
#else
#define RTPOSE_TSTAMP(slot)
#endif

// per-tile stamps of the persistent 3x3 Winograd kernel (tools/timeline_w3.py)
#ifdef RTPOSE_EXP_TIMELINE3
#define RTPOSE_TSTAMP3(ti, slot) \
  if (A.dbg && threadIdx.x == 0 && (ti) < 6) A.dbg[((size_t)blockIdx.x * 6 + (ti)) * 8 + (slot)] = __builtin_amdgcn_s_memtime()
#else
#define RTPOSE_TSTAMP3(ti, slot)
#endif

// per-chunk stamps of every wave of the F(4x4,3x3) kernel over a block's first tile (tools/timeline_w4.py): slot 0 = first
// MFMA of the chunk may issue (after the barrier), 1 = last MFMA issued (before the barrier), chunk 63 = epilogue begin / end
#ifdef RTPOSE_EXP_TIMELINE4
#define RTPOSE_TSTAMP4(chunk, slot)                                                                           \
  if (A.dbg && (threadIdx.x & 63) == 0 && (chunk) < 64)                                                        \
  A.dbg[(((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + (chunk)) * 2 + (slot)] = __builtin_amdgcn_s_memtime()
#else
#define RTPOSE_TSTAMP4(chunk, slot)
#endif

// fp32 1x1 convs: CK-channel sub-chunks per LDS buffer (4: -20 % on the 1x1 layers)
#ifndef RTPOSE_EXP_TB1X1
#define RTPOSE_EXP_TB1X1 1
#endif
// bf16: depth of the halo staging ring / of the weight prefetch ring of the 2 x 2 wave form
#ifndef RTPOSE_EXP_HD
#define RTPOSE_EXP_HD 3
#endif
#ifndef RTPOSE_EXP_RB2
#define RTPOSE_EXP_RB2 4
#endif

// Winograd kernels: accumulator registers per lane that go through the output transform + stores (16; 1 = timing-only
// variant that shows what the epilogue costs)
#ifndef RTPOSE_EXP_W_EPI
#define RTPOSE_EXP_W_EPI 16
#endif
// 3x3 Winograd kernel: rotation of the wtile slots per channel group in LDS (8 / CG: write-conflict-free;
// round 2 used 16 / CG)
#ifdef RTPOSE_EXP_W3_SWOLD
#define RTPOSE_EXP_W3_SW(CG) (16 / (CG))
#else
#define RTPOSE_EXP_W3_SW(CG) (8 / (CG))
#endif
// 7x7 Winograd kernel, 8-wave form: B register sets (3 or 7; prefetch distance = min(RTPOSE_EXP_W7_PF, sets - 1))
#ifndef RTPOSE_EXP_W7_FS2SETS
#define RTPOSE_EXP_W7_FS2SETS 7
#endif
// 7x7 Winograd kernel, 8-wave form: 1 = sibling waves share the input transform of an item (half the groups each)
#ifndef RTPOSE_EXP_W7_XSPLIT
#define RTPOSE_EXP_W7_XSPLIT 0
#endif
// 7x7 Winograd kernel, 8-wave form: issue priority (s_setprio) of the transforming waves: 0 none, 1 higher, 2 lower
#ifndef RTPOSE_EXP_W7_PRIO
#define RTPOSE_EXP_W7_PRIO 0
#endif
// 7x7 Winograd kernel: transform items of a row ordered channel-group-major (1) so that the 8 lanes of a
// ds_write_b128 group hit 8 distinct 16-byte slots; 0 = round 2's (gx, cg) order (2-way conflicts on every write)
#ifndef RTPOSE_EXP_W7_CGMAJOR
#define RTPOSE_EXP_W7_CGMAJOR 1
#endif
// F(4,7) kernel: weight prefetch distance in (ky, frequency pair) steps (<= 4: 5 register sets; 2: +5 %, 3: +0.7 %)
#ifndef RTPOSE_EXP_W7_PF
#define RTPOSE_EXP_W7_PF 4
#endif

// F(4,7) kernel: which parts of the next chunk's input transform run in the multiply loop (1: the VALU groups +
// LDS writes, 2: the segment loads)
// segment loads per step (1: 0.719 -> 0.695 ms per 128 -> 128 layer against 2; the loads touch 32 cache lines
// each and all four waves issue them in the same steps, 5 per step 0.718)
#ifndef RTPOSE_EXP_W7_LPS
#define RTPOSE_EXP_W7_LPS 1
#endif
#ifndef RTPOSE_EXP_W7_TMASK
#define RTPOSE_EXP_W7_TMASK 3
#endif

// fp32: which B register (k-group) is fetched after MFMA pair n (-1 = none); GB is the kernel's
#ifdef RTPOSE_EXP_BSPREAD
#define RTPOSE_EXP_BSLOT(n) (((n) % 2 == 0 && (n) / 2 < GB) ? (n) / 2 : -1)
#else
#define RTPOSE_EXP_BSLOT(n) (((n) < GB) ? (n) : -1)
#endif
#ifdef RTPOSE_EXP_HALF_B_ON
#define RTPOSE_EXP_HALF_B 1
#else
#define RTPOSE_EXP_HALF_B 0
#endif
// drop one load stream at a time: weights (B), LDS fragment reads (A), next-chunk halo staging
#ifdef RTPOSE_EXP_NO_B
#define RTPOSE_EXP_B(load, keep) (keep)
#else
#define RTPOSE_EXP_B(load, keep) (load)
#endif
#ifdef RTPOSE_EXP_NO_A
#define RTPOSE_EXP_A(load, keep) (keep)
#else
#define RTPOSE_EXP_A(load, keep) (load)
#endif
// bf16: timing-only - every chunk re-stages chunk 0 of the tile's own halo (L2-resident after the first pass): what does
// the LATENCY of the staging loads cost, as opposed to their issue slots and LDS writes?
#ifdef RTPOSE_EXP_STAGE_NEAR
#define RTPOSE_EXP_STAGE_SRC(next, own) (own)
#else
#define RTPOSE_EXP_STAGE_SRC(next, own) (next)
#endif
#ifdef RTPOSE_EXP_NO_STAGE
#define RTPOSE_EXP_STAGE 0
#else
#define RTPOSE_EXP_STAGE 1
#endif

// tools/exp/conv_wino16.hip (experiment record): B ring entries / prefetch distance (frequencies)
#ifndef RTPOSE_W16_NB
#define RTPOSE_W16_NB 8
#endif
#ifndef RTPOSE_W16_PF
#define RTPOSE_W16_PF 6
#endif

// conv_wino4.hip: first pair step of a chunk's patch loads
#ifndef RTPOSE_EXP_W4_L0
#define RTPOSE_EXP_W4_L0 6  // (of 9 steps; two loads per step)
#endif
// conv_wino4.hip: persistent blocks of one m tile on one XCD (1) or spread over the XCDs (0)
#ifndef RTPOSE_EXP_W4_XCDMAP
#define RTPOSE_EXP_W4_XCDMAP 1
#endif
// conv_wino7.hip, 8-wave form: 1 = the non-transforming waves issue the segment loads too, through a zero-extent descriptor
// (exact vmcnt bookkeeping; what gave the F(4x4,3x3) kernel 6 % measured 0 % here: 14.18 vs 14.12..14.19 ms of 7x7 time)
#ifndef RTPOSE_EXP_W7_NULLDESC
#define RTPOSE_EXP_W7_NULLDESC 0
#endif
// conv_wino4.hip: s_setprio scheme of the two waves of a SIMD (0 none, 1 alternate per step, 2 younger wave high, 3 alternate per chunk)
#ifndef RTPOSE_EXP_W4_PRIO
#define RTPOSE_EXP_W4_PRIO 0
#endif
// conv_wino4.hip: A fragments in a ring of three (requested 8 MFMAs ahead) instead of two (4 ahead)
#ifndef RTPOSE_EXP_W4_A3
#define RTPOSE_EXP_W4_A3 0
#endif
