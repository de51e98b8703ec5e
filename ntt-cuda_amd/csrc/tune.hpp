// tune.hpp -- the tuning constants of the single-pass kernels (ntt_core.cuh, kernels_fast_impl.cuh, kernels_lat.cuh), in ONE place.
//
// Rounds 1-5 carried ~55 compile-time experiment switches inside the kernel sources (ablations that produced wrong results,
// in-kernel time stamps, alternative code paths); round 6 froze the kernels and removed all of them.  What is
// left is this struct: plain values the kernels read as template arguments / constants.  The library always builds with the values
// below.  A measurement build (tools/build_kbench.sh) may substitute another struct of the same shape by defining
// MI355NTT_TUNE_HEADER to the path of a header that defines `mi355ntt::Tune`; nothing else in csrc/ looks at a macro to change a
// kernel.  The ablation / stamp builds behind profiles/r01..r05_* are reproducible from the round-5 tree (commit 3b06c3b).
//
// Each value names the profile that chose it.
#pragma once

#ifdef MI355NTT_TUNE_HEADER
#include MI355NTT_TUNE_HEADER
#else
namespace mi355ntt {
struct Tune {
    // ---- cache policy (aux bits of the buffer instructions: 1 = sc0, 2 = nt, 16 = sc1) ----
    static constexpr int kStreamAuxLd = 0;      // coalesced polynomial loads: default policy (nt: forward +-0, inverse -4..5 % at 4096 polynomials; profiles/r01_memory_system_experiments.txt)
    static constexpr int kStreamAuxSt = 0;      // forward result stores: default (nt stores cost 3-7 % everywhere)
    static constexpr int kRowsAuxLd = 0;        // 16-byte row loads of the inverse / fused kernels, default policy ...
    static constexpr int kInv15AuxLd = 2;       // ... and in k_inverse15 launches that stream (>= 4096 polynomials, kStreamLoads): nt, inverse alone -1..2 %, pairs +-0 (profiles/r05_streaming_overlap.txt, section 9)
    static constexpr int kMul15BAuxLd = 2;      // k_polymul15's own second operands beyond 512 polynomials: nt, -1..5 % per product (same file, section 10)
    static constexpr int kInv15AuxSt = 17;      // k_inverse15 / k_polymul15 result stores written through at system scope: no write-back at kernel end, inverse launches back to back -3.9 % (profiles/r03_structural_experiments.txt, batch C)
    static constexpr int kInvAuxSt = 17;        // the same for n = 2^11 .. 2^14 (small, never worse)
    // ---- wave priorities per phase of the n = 2^15 kernels (s_setprio; kPsplitX = scheduling group from which kPrioXB applies, -1 = none)
    // the phase right after the workgroup-wide exchange highest, the round that feeds it lowest: +3..7 % (profiles/r02_priority_and_noise.txt)
    static constexpr int kPrioR1 = 0, kPsplitR1 = -1, kPrioR1B = 0;
    static constexpr int kPrioR2 = 3, kPsplitR2 = -1, kPrioR2B = 0;
    static constexpr int kPrioR3 = 2, kPsplitR3 = 12, kPrioR3B = 1;
    static constexpr int kPrioI1 = 3, kPsplitI1 = -1, kPrioI1B = 0;
    static constexpr int kPrioI2 = 0, kPsplitI2 = -1, kPrioI2B = 0;
    static constexpr int kPrioI3 = 2, kPsplitI3 = -1, kPrioI3B = 0;
    // ---- start stagger of the persistent workgroups: 8 phase groups, units x 2048 cycles apart (one polynomial per workgroup / several)
    // profiles/r02_stagger_retune.txt; fused kernel: profiles/r05_streaming_overlap.txt, section 16
    static constexpr int kStaggerFwd = 1, kStaggerFwdMulti = 2;
    static constexpr int kStaggerInv = 0, kStaggerInvMulti = 2;
    static constexpr int kStaggerMul = 0, kStaggerMulMulti = 2;
    // ---- shapes ----
    static constexpr int kSchedGroup = 4;       // butterflies per scheduling group (a fence per group bounds the live twiddle set)
    static constexpr int kRingGroupB0 = 4, kRingDepthB0 = 2;      // twiddle ring of the bit-0 round at n = 2^15
    static constexpr int kTwoPhaseMinLogN = 13; // half-size LDS image (two-phase exchange) from n = 2^13 up
    static constexpr bool kFwdLoadPair16 = true;      // forward loads issued in consumption order (0, 16, 1, 17, ...): +0.8 %
    static constexpr bool kInvDescending = true;      // k_inverse15 walks the batch from the last polynomial down: +2.5 % pairs at 1024 (profiles/r02_walk_order_and_store_policy.txt)
};
}  // namespace mi355ntt
#endif
