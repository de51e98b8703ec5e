// hostparams.hpp -- host-only parameter derivation for the NTT engine (no HIP).
// Replaces the reference's helper.h / parameter.h bootstrap; see include/mi355ntt.h for citations.
#pragma once
#include <cstdint>
#include <vector>

namespace mi355ntt {

using u64 = unsigned long long;
using u128 = unsigned __int128;

unsigned bit_length(u64 q);
u64 barrett_mu(u64 q, unsigned k);
u64 mulmod(u64 a, u64 b, u64 m);
u64 modpow(u64 a, u64 e, u64 m);
u64 modinv(u64 a, u64 q);
u64 bit_reverse(u64 a, int bits);
// floor(w * 2^64 / q): the Shoup companion of a constant multiplier w < q
u64 shoup(u64 w, u64 q);
void fill_table(u64 root, u64 q, unsigned n, u64* tab);

// Does the reference's Barrett (Algorithm 7, ONE conditional subtraction: ntt_60bit.cuh:44-61) return the canonical
// residue for every product of two canonical operands?  See hostparams.cpp.
bool barrett_single_subtraction_exact(u64 q, unsigned k, u64 mu);
bool barrett_exact_for_operand_q(u64 q, unsigned k, u64 mu);      // ... with one operand equal to q itself

// Everything the device kernels need for one prime.
struct PrimeParams {
    u64 q, psi, psiinv, mu, ninv;
    unsigned k;
    bool barrett_exact;   // barrett_single_subtraction_exact(q, k, mu)
};

// Validates (q, psi) for ring degree n; returns 0 or a negative MI355NTT_E* code.
int derive_prime(unsigned n, u64 q, u64 psi, PrimeParams* out);

}  // namespace mi355ntt
