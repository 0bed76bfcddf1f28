// kernels_stats_layout.h -- LDS layout constants of the pipelined statistics GEMM, shared by kernels_stats.h and the
// statistics workgroups of the fused sweep + statistics kernel (kernels_fused.h).
#pragma once
#ifndef ST_RB
#define ST_RB 32
#endif
#define ST_CS 33            // row slots per buffer per column (32 + 1 pad)
#define ST_CC (2 * ST_CS + 1)  // column stride (both buffers + 1 pad): 67
// LDS banking of the two operand tiles (round 6; tools/probe/lds_probe.hip, profiles/r06a_lds_probe_pmc.txt).  A wave's
// ds_read_b64 is served in two groups of 32 lanes on 64 four-byte banks, i.e. the 32 eight-byte words of lanes
// (li = 0..15, lg = 0, 1) -- and of (li, lg = 2, 3) -- must fall on 32 different bank pairs.  Rounds 1-5 laid both
// tiles out for 16-lane groups (row stride of the q tile Kp + 1, the four k-rows of a k-step eight row slots apart
// in a column): conflict-free within one lg, but lg = 0 and 1 collided on 8 (A operand) / 15 (B operand) of the 32
// pairs and every operand read took twice its LDS cycles (SQ_LDS_BANK_CONFLICT = SQ_LDS_IDX_ACTIVE / 2).  Now:
//   q tile: row stride = 16 (mod 32) words, so lg = 1 lands on the pairs lg = 0 leaves free;
//   A tile: the row slot of (lg, ks) is 16 (lg & 1) + 8 (lg >> 1) + ks -- with an odd column stride (67 / 101) the
//   sixteen columns of a feature tile cover sixteen pairs m .. m + 15 in multiples of the stride, and the same columns
//   sixteen slots further the other sixteen.
#ifdef SVIHMM_AB_STATS_OLD      // (A/B builds only: the layout of rounds 1-5)
#define ST_QS(KP) ((KP) + 1)
#define ST_SLOT(LG) (8 * (LG))
#else
#define ST_QS(KP) ((((KP) + 15) / 32) * 32 + 16)
#define ST_SLOT(LG) (16 * ((LG) & 1) + 8 * ((LG) >> 1))
#endif
// the 128 x 64 transition blocks of wide models (TRONLY, two q[prev] groups per workgroup) fill the LDS with the
// three-buffer loop: their q tile keeps the Kp + 1 stride (161 KB; 172 KB with ST_QS)
#define ST_QS_TR(KP, MT) ((MT) >= 2 ? (KP) + 1 : ST_QS(KP))

// the barrier-free three-buffer loop at XK >= 5 (48 <= D: 131 columns at D = 64) has 161 KB of LDS with the Kp + 1 stride
// and would not fit with ST_QS (172 KB) -- it would fall back to the double-buffered kernel: 20.9 instead of 19.2 ms on
// configs[4] (measured, profiles/r06z_c5_kernel_stats.txt against r05j's).  Those shapes keep the old stride.
#define ST_QS3(KP, XK) ((XK) >= 5 ? (KP) + 1 : ST_QS(KP))
