"""Parameter sets shared by the tests; every (q, psi) cites its origin."""
import hashlib

import numpy as np

# reference getParams, BFV_Scheme/parameter.h:31-79: n -> (q, psi, psiinv, ninv, q_bit)
REF_PARAMS = {
    2048: (137438691329, 22157790, 88431458764, 137371582593, 37),
    4096: (33538049, 2386, 26102329, 33529861, 25),
    8192: (8796092858369, 1734247217, 5727406356888, 8795019116565, 43),
    16384: (281474976546817, 23720796222, 129310633907832, 281457796677643, 48),
    32768: (36028797017456641, 1155186985540, 31335194304461613, 36027697505828911, 55),
}
# the commented-out 58-bit set, parameter.h:43-47
REF_PARAMS_4096_58BIT = (288230376135196673, 60193018759093, 236271020333049746, 288160007391023041, 58)

# BASELINE configs 2-4: the four largest 60-bit primes = 1 mod 2^16 with their minimal primitive 2n-th roots
Q60 = [1152921504606584833, 1152921504598720513, 1152921504597016577, 1152921504595968001]
PSI60 = [4443670208963, 100545759574150, 31693996050849, 88651361085495]

# BASELINE configs[4] (BFV, 4 x 60-bit RNS): the special prime encryption drops = next 60-bit prime = 1 mod 2^16 below Q60[3]
Q60_SPECIAL, PSI60_SPECIAL = 1152921504595640321, 9679305630873

# primes far from a power of two (q = 1 mod 2^16, minimal primitive 2^16-th roots): the kernels pick their partial
# reduction and butterfly form by headroom class (64 - bit length) and by "near 2^k or not" -- these reach the general forms
GENERAL_PRIMES = {57: (93674872251744257, 1408945640707), 59: (536108499642941441, 7338562720162),
                  60: (818574268271493121, 19054799908346), 62: (3827699395296821249, 61619156825551)}

# general 61-bit primes (= 1 mod 2^17, far from 2^61 or with 2^61 - q >= 2^24): class 3 with the conditional-subtract inverse since
# round 5 (before: class 2, exact quotients): q -> {n: a primitive 2n-th root}
GENERAL61 = {1609682519816667137: {32768: 1058825331674317735, 65536: 168149748110227714},
             2305843009195868161: {32768: 323510840180264020, 65536: 1454513390878286822}}

# the reference's first four 55-bit demo primes, BFV_Scheme/demo.cu:35-36 (entries 1..4 of the 16-prime set)
Q55 = [36028797017456641, 36028797014704129, 36028797014573057, 36028797014376449]
PSI55 = [1155186985540, 631260524634, 1526647220035, 455957817523]

# 61/62-bit moduli to exercise the top of the supported range (q < 2^62); gamma is the reference's
# decryption auxiliary modulus (demo.cu:93), a 61-bit prime = 1 mod 2^16
GAMMA61 = 2305843009213683713

# SURVEY.md 4.2 golden digests: (n, q, psi) -> (sha256(psi table), sha256(NTT(a)), sha256(INTT(NTT(a) . NTT(b))))
# a = splitmix64(seed 1) mod q, b = splitmix64(seed 2) mod q
GOLDEN_DIGESTS = {
    (4096, 288230376135196673, 60193018759093): (
        "f867e7464fc7ed4f85051e41148bcef14ef6c8a45b5c287c71aacc037b0772f9",
        "d2c6d4920f2127a10134c124c893581354ae236e5655bc65796968c1a5455394",
        "c4ab96ced1973bf6a5f72b625197153308dbf2e9bd45341478fb60c989beb83f"),
    (4096, 33538049, 2386): (
        "3b3a78d606c926320ceb8557ef0a52a28600005703031c4de63e67c9a4d41377",
        "bd98ec4349920eb70ad8daf2a6ba53622a7800a0466e4b51f38d98d26e57d8ad",
        "c8331dcb1b4cc3b2e11e35a514b04410bf0d6609f5dec191e04a8883eb92344f"),
    (32768, 36028797017456641, 1155186985540): (
        "cfbe21cc6339c8cc211506f1fd94bd0dae29f906043eb907c2d6bad14439649b",
        "b6c635bbbb31fe071261fb8b66ae5811e81b3d37fea87196fd571d71db4c5960",
        "7831d19a8e66efdfaff28dbd02a84ae1c425982ce0192e61e8ae295dde1d41cd"),
    (32768, 1152921504606584833, 4443670208963): (
        "163c8367a7df57476b25b1b498de63f53e5ff2ac197ccecc604caccf756f4b89",
        "17b03fe2badfccaad9e252c5208ba2e709e97251bde2c0e9c9372c99873df6bb",
        "6c6350c5054f97068de1a2e3fa0cf04503abae6323b83ef1e59ffcf38087cb9b"),
}

# KAT-1 (decryption_test.cu:348,355): digests of c1 after forward batch / barrett_batch / inverse batch
KAT1_STAGE_DIGESTS = (
    "355d443406d4ad2b46cbceab8b5728e9e8a035971fb0a4f4bd5eaa9bd974275a",
    "3820e8f1c17f7af44decb0f4ecae6bcb6991eef13fbabcafcfcb38d57f08a953",
    "4496cba955c64d707885b7eed6a09b148403c9e1aa39fa21c7a097bdc87f9ac7",
)


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).astype("<u8").tobytes()).hexdigest()

# Edge-of-range primes (= 1 mod 2^17), found by search; value -> {n: primitive 2n-th root}
EDGE_PRIMES = {
    62: (4611686018425815041, {65536: 2824515048472102463, 32768: 852445115902909273, 4096: 4411335539154205079}),
    61: (2305843009211596801, {65536: 1681162619342215248, 32768: 572539251920920660, 4096: 700439432845261874}),
    59: (576460752300015617, {65536: 296969298802020438, 32768: 178988506022562004, 4096: 379167746563934504}),
    30: (1073479681, {65536: 1070907127, 32768: 31849551, 4096: 371836615}),
}

# six 62-bit primes = 1 mod 2^17 just below 2^62 with primitive 8192-th roots (n = 4096), found by search (sympy.isprime, x^((q-1)/2n)
# with w^n = -1), and the largest 40-bit prime: moduli WIDER than gamma, where a Barrett product mod gamma is no longer below 2 gamma
# (the lazy sum of k_decrypt_round must fall back to the reference's per-term reduction)
Q62_N4096 = [(4611686018425815041, 4411335539154205079), (4611686018423062529, 1987687080756694092), (4611686018422669313, 3419936153954982955),
             (4611686018416115713, 1744709284562504997), (4611686018408120321, 1965406977792563506), (4611686018406940673, 1210959489459782225)]
GAMMA40 = (1 << 40) - 87

# Barrett-INEXACT primes (mi355ntt_barrett_is_exact == 0: the reference's single-subtraction Barrett returns q + r for some operand
# pairs), = 1 mod 2^17, found by search (sympy.isprime + the host predicate): bits -> (q, {n: primitive 2n-th root}).  Such primes sit
# in the upper part of [2^(k-1), 2^k) and far enough from 2^k that frac(2^(2k) / q) is large.
INEXACT_PRIMES = {
    34: (16717447169, {2048: 716892639, 4096: 10780603935, 8192: 9801559072, 16384: 14982407107, 32768: 14234614852, 65536: 3867875371}),
    36: (66607251457, {2048: 399654466, 4096: 66172355353, 8192: 62832410156, 16384: 32337211720, 32768: 55503464023, 65536: 28050426572}),
    50: (1090484111147009, {2048: 849604439456216, 4096: 275136633019418, 8192: 557074390833951, 16384: 863481998425754, 32768: 110572600602772,
                            65536: 553978784588634}),
    60: (1137833256315125761, {2048: 735943513308022933, 4096: 1024597489263776217, 8192: 418497743808556150, 16384: 206061080305443116,
                               32768: 448230823712243253, 65536: 294981370794764583}),
    61: (2248020882338086913, {2048: 163138043264649591, 4096: 1863497830520578066, 8192: 1290552877515215379, 16384: 896052032404846333,
                               32768: 423556294508823369, 65536: 1593308822622025087}),
}
# Barrett-exact primes of similar size (same search, predicate true) to sit next to them in one context
EXACT_NEIGHBOURS = {
    36: (65803911169, {2048: 46725621650, 4096: 9944776363, 8192: 26729495459, 16384: 30672130432, 32768: 2091551950}),
    60: (1137354327060774913, {2048: 168727745372988259, 4096: 1048799164746730205, 8192: 491669866052357047, 16384: 234609454235331695,
                               32768: 492103990670350833}),
}
