// bfv_demo.cpp -- the reference's end-to-end program (BFV_Scheme/demo.cu) rebuilt on the compat header: parameter set,
// buffers, keygen_rns -> encryption_rns -> decryption_rns from the keystream, event timing and the correctness check
// of demo.cu:275-320.  What changed against demo.cu: the bootstrap block (demo.cu:62-272) is one mi355ntt_bfv_create
// call, the drivers take the object instead of twenty parameter arrays, cuda* -> hip*.
// Build (tests/test_cpp_compat.py): hipcc -std=c++17 tests/cpp/bfv_demo.cpp -L ntt-cuda_amd -lmi355ntt -o tests/cpp/bfv_demo
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../ntt-cuda_amd/compat/bfv_launch.hpp"

using namespace mi355;

#define HIPCK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return 2; } } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) { printf("%s -> %s\n", #x, mi355ntt_strerror(rc_)); return 3; } } while (0)

int main(int argc, char** argv)
{
    const unsigned n = 1024 * 32;                                                       // demo.cu:26
    const unsigned long long t = 1024;                                                  // demo.cu:28
    // 32k 16q, demo.cu:35-36
    std::vector<unsigned long long> q_array = {18014398506729473, 36028797017456641, 36028797014704129, 36028797014573057,
        36028797014376449, 36028797013327873, 36028797013000193, 36028797012606977, 36028797010444289, 36028797009985537,
        36028797005856769, 36028797005529089, 36028797005135873, 36028797003694081, 36028797003563009, 36028797001138177};
    std::vector<unsigned long long> psi_roots = {58232959302, 1155186985540, 631260524634, 1526647220035, 455957817523,
        1650884166641, 10316746886, 768741990072, 3911086673862, 5947090524825, 47595902954, 2691682578057, 3903338373,
        235185854118, 1769787302793, 3151164484090};
    unsigned primes = argc > 1 ? (unsigned)atoi(argv[1]) : 16;                          // q_amount + 1 of demo.cu
    if (primes < 2 || primes > 16) { printf("2..16 primes\n"); return 2; }
    const unsigned q_amount = primes - 1;
    const unsigned long long gamma = 2305843009213683713ULL;                            // demo.cu:93

    mi355ntt_bfv* bfv;                                                                  // demo.cu:62-272
    RC(mi355ntt_bfv_create(&bfv, n, primes, q_array.data(), psi_roots.data(), t, gamma, 0, 0));

    unsigned char* in;                                                                  // demo.cu:142-143 (sized for what keygen_rns writes)
    HIPCK(hipMalloc(&in, mi355ntt_bfv_keygen_random_bytes(bfv)));
    unsigned long long *secret_key, *public_key, *temp, *c, *e, *m_poly_device;
    const size_t poly = sizeof(unsigned long long) * n;
    HIPCK(hipMalloc(&secret_key, poly * primes));                                       // demo.cu:145-160
    HIPCK(hipMalloc(&public_key, poly * primes * 2));
    HIPCK(hipMalloc(&temp, poly * primes));
    HIPCK(hipMalloc(&c, poly * primes * 2));
    HIPCK(hipMalloc(&e, poly * primes * 2));
    HIPCK(hipMalloc(&m_poly_device, poly));
    std::vector<unsigned long long> m_poly(n);
    for (unsigned i = 0; i < n; i++) m_poly[i] = (i * 2654435761u) % t;                 // demo.cu fills it with rand() % t
    HIPCK(hipMemcpy(m_poly_device, m_poly.data(), poly, hipMemcpyHostToDevice));

    hipEvent_t start, stop;
    HIPCK(hipEventCreate(&start));
    HIPCK(hipEventCreate(&stop));
    float keygen = 0, enc = 0, dec = 0;
    for (int pass = 0; pass < 2; pass++) {                                              // pass 0 warms the clocks and caches up
        HIPCK(hipEventRecord(start));                                                   // KEYGEN, demo.cu:275-280
        RC(keygen_rns(bfv, in, secret_key, public_key, temp, nullptr));
        HIPCK(hipEventRecord(stop));
        HIPCK(hipEventSynchronize(stop));
        HIPCK(hipEventElapsedTime(&keygen, start, stop));
        HIPCK(hipEventRecord(start));                                                   // ENCRYPTION, demo.cu:282-288
        RC(encryption_rns(bfv, c, public_key, in, e, m_poly_device, nullptr, /*nonce*/ 1));
        HIPCK(hipEventRecord(stop));
        HIPCK(hipEventSynchronize(stop));
        HIPCK(hipEventElapsedTime(&enc, start, stop));
        HIPCK(hipEventRecord(start));                                                   // DECRYPTION, demo.cu:290-297
        RC(decryption_rns(bfv, c, secret_key, nullptr));
        HIPCK(hipEventRecord(stop));
        HIPCK(hipEventSynchronize(stop));
        HIPCK(hipEventElapsedTime(&dec, start, stop));
    }
    std::vector<unsigned long long> decrypted(n);                                       // demo.cu:299-311
    HIPCK(hipMemcpy(decrypted.data(), c + (size_t)n * (q_amount - 1), poly, hipMemcpyDeviceToHost));
    bool correct = true;
    for (unsigned i = 0; i < n; i++)
        if (m_poly[i] != decrypted[i]) { correct = false; break; }
    printf("n = %u, %u primes (log q = %u): keygen %.1f us, encryption %.1f us, decryption %.1f us\n", n, primes,
           54 + 55 * (primes - 1), keygen * 1e3f, enc * 1e3f, dec * 1e3f);
    printf(correct ? "Decryption is correct\n" : "Decryption is WRONG\n");

    // ---- many ciphertexts per call (argv[2] = count): sampled per ciphertext as above, laid out [2][count][primes][n],
    // encrypted and decrypted with one call each; every message must come back ----
    const unsigned count = argc > 2 ? (unsigned)atoi(argv[2]) : 0;
    if (correct && count > 0) {
        unsigned long long *cb, *eb, *mb;
        const size_t half = poly * primes;                                              // one component of one ciphertext
        HIPCK(hipMalloc(&cb, 2 * half * count));
        HIPCK(hipMalloc(&eb, 2 * half * count));
        HIPCK(hipMalloc(&mb, poly * count));
        std::vector<unsigned long long> msgs((size_t)n * count);
        for (size_t i = 0; i < msgs.size(); i++) msgs[i] = (i * 40503u + (i >> 7)) % t;
        HIPCK(hipMemcpy(mb, msgs.data(), poly * count, hipMemcpyHostToDevice));
        const unsigned char key[32] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
        for (unsigned z = 0; z < count; z++) {                                          // fresh keystream -> u | u and e0 | e1 of ciphertext z
            RC(mi355ntt_salsa20_keystream(in, mi355ntt_bfv_encrypt_random_bytes(bfv), key, 100 + z, nullptr));
            RC(mi355ntt_bfv_sample_encrypt(bfv, in, c, e, nullptr));
            for (int h = 0; h < 2; h++) {
                HIPCK(hipMemcpyAsync((char*)cb + (h * (size_t)count + z) * half, (char*)c + h * half, half, hipMemcpyDeviceToDevice, nullptr));
                HIPCK(hipMemcpyAsync((char*)eb + (h * (size_t)count + z) * half, (char*)e + h * half, half, hipMemcpyDeviceToDevice, nullptr));
            }
        }
        float encb = 0, decb = 0;
        HIPCK(hipEventRecord(start));
        RC(encryption_rns_batch(bfv, cb, public_key, eb, mb, count, nullptr));
        HIPCK(hipEventRecord(stop));
        HIPCK(hipEventSynchronize(stop));
        HIPCK(hipEventElapsedTime(&encb, start, stop));
        HIPCK(hipEventRecord(start));
        RC(decryption_rns_batch(bfv, cb, secret_key, count, nullptr));
        HIPCK(hipEventRecord(stop));
        HIPCK(hipEventSynchronize(stop));
        HIPCK(hipEventElapsedTime(&decb, start, stop));
        std::vector<unsigned long long> back(n);
        for (unsigned z = 0; z < count && correct; z++) {
            HIPCK(hipMemcpy(back.data(), cb + ((size_t)z * primes + primes - 2) * n, poly, hipMemcpyDeviceToHost));
            for (unsigned i = 0; i < n; i++)
                if (back[i] != msgs[(size_t)z * n + i]) { correct = false; break; }
        }
        printf("%u ciphertexts per call: encryption %.1f us, decryption %.1f us (%.2f / %.2f us per ciphertext)\n", count, encb * 1e3f,
               decb * 1e3f, encb * 1e3f / count, decb * 1e3f / count);
        printf(correct ? "Batched decryption is correct\n" : "Batched decryption is WRONG\n");
    }
    mi355ntt_bfv_destroy(bfv);
    return correct ? 0 : 1;
}
