// shard.cpp -- a batch over several devices from ONE process (include/mi355ntt.h, "Multi-GPU").
//
// The reference is single-GPU.  Its batch kernels never let polynomials interact -- blockIdx.y selects the data offset y*n and the
// modulus y % division and nothing else (ntt_60bit.cuh:391,404,422) -- so a batch shards into contiguous ranges of WHOLE polynomials
// whose starts are multiples of `division` (polynomial y keeps prime y % division inside its shard), with one replicated context per
// device and no collective on the data path.  This file is the single-process form of that rule (the one-process-per-GPU form over
// torch.distributed / RCCL lives in ntt_cuda_amd/shard.py): device-resident shards transformed concurrently, and a root-resident batch
// dealt out by peer copies over xGMI, transformed and copied back, pipelined per device on three streams.  It is written against the
// public C ABI only (mi355ntt_forward_batch ...): a layer above the engine, not part of it.
#include "../../include/mi355ntt.h"

#include <hip/hip_runtime.h>

#include <new>
#include <vector>

#include "device_scope.hpp"

namespace mi355ntt {
void record_hip_error(int e);      // capi.cpp: what mi355ntt_last_hip_error() reports
}
using mi355ntt::DeviceScope;

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e__ = (expr);                       \
        if (e__ != hipSuccess) {                       \
            mi355ntt::record_hip_error((int)e__);      \
            return MI355NTT_EHIP;                      \
        }                                              \
    } while (0)
#define RC_TRY(expr)                \
    do {                            \
        int rc__ = (expr);          \
        if (rc__) return rc__;      \
    } while (0)

namespace {
constexpr int kRing = 3;       // staging buffers per device: piece k + 1 arriving, piece k in the kernels, piece k - 1 on its way back

struct Lane {                  // one shard = one context on one device
    const mi355ntt_ctx* ctx = nullptr;
    int device = 0;
    hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
    hipEvent_t ev_in[kRing] = {}, ev_cmp[kRing] = {}, ev_out[kRing] = {}, ev_done = nullptr;
    mi355ntt_u64* stage[kRing] = {};
};
}  // namespace

struct mi355ntt_shards {
    unsigned world = 0, n = 0, num_primes = 0, piece_polys = 0;
    std::vector<Lane> lane;
    hipEvent_t ev_start = nullptr;     // on the root device
};

static int run_op(const mi355ntt_ctx* c, int op, mi355ntt_u64* d, const mi355ntt_u64* bhat, unsigned count, unsigned division, hipStream_t s)
{
    switch (op) {
    case MI355NTT_OP_FORWARD: return mi355ntt_forward_batch(c, d, count, division, s);
    case MI355NTT_OP_INVERSE: return mi355ntt_inverse_batch(c, d, count, division, s);
    case MI355NTT_OP_FORWARD_INVERSE:
        RC_TRY(mi355ntt_forward_batch(c, d, count, division, s));
        return mi355ntt_inverse_batch(c, d, count, division, s);
    case MI355NTT_OP_POLYMUL: return bhat ? mi355ntt_polymul_batch(c, d, bhat, count, division, s) : MI355NTT_EINVAL;
    default: return MI355NTT_EINVAL;
    }
}

// Error path of the two fork / join calls: whatever was enqueued on the lanes' streams before the failure is allowed to finish (host
// synchronisation -- this is not the fast path), so that an error code never comes back while another device still writes the
// caller's buffers.
static void drain_lanes(mi355ntt_shards* s)
{
    for (Lane& l : s->lane) {
        DeviceScope scope(l.device);
        if (scope.err != hipSuccess) continue;
        if (l.s_in) (void)hipStreamSynchronize(l.s_in);
        if (l.s_cmp) (void)hipStreamSynchronize(l.s_cmp);
        if (l.s_out) (void)hipStreamSynchronize(l.s_out);
    }
    (void)hipGetLastError();
}

// pieces of a shard of `count` polynomials: at most `chunks`, at most piece_polys each, starts at multiples of `division`
static void piece_plan(unsigned count, unsigned division, unsigned chunks, unsigned piece_polys, std::vector<std::pair<unsigned, unsigned>>* out)
{
    out->clear();
    if (!count) return;
    const unsigned groups = count / division, tail = count % division;
    unsigned per = groups ? (groups + chunks - 1) / chunks : 0;               // groups per piece
    const unsigned cap = piece_polys / division;                               // (>= 1: checked by the caller)
    if (per > cap) per = cap;
    if (per == 0) per = 1;
    unsigned g = 0;
    while (g < groups) {
        const unsigned gc = groups - g < per ? groups - g : per;
        out->push_back({g * division, gc * division});
        g += gc;
    }
    if (tail) {                                                                // a ragged tail rides with the last piece when it fits
        if (!out->empty() && out->back().second + tail <= piece_polys) out->back().second += tail;
        else out->push_back({groups * division, tail});
    }
}

extern "C" {

/* contiguous ranges of whole polynomials, starts at multiples of `division`, groups dealt as evenly as possible, a ragged tail
 * (num % division polynomials) with the last rank: the rule of ntt_cuda_amd/shard.py shard_range, word for word */
int mi355ntt_shard_range(unsigned num, unsigned division, unsigned rank, unsigned world, unsigned* first, unsigned* count)
{
    if (division == 0 || world == 0 || rank >= world) return MI355NTT_EINVAL;
    const unsigned groups = num / division, tail = num % division;
    const unsigned base = groups / world, extra = groups % world;
    const unsigned g0 = rank * base + (rank < extra ? rank : extra);
    const unsigned gcount = base + (rank < extra ? 1u : 0u);
    if (first) *first = g0 * division;
    if (count) *count = gcount * division + (rank == world - 1 ? tail : 0u);
    return MI355NTT_OK;
}

int mi355ntt_shards_destroy(mi355ntt_shards* s)
{
    if (!s) return MI355NTT_OK;
    for (Lane& l : s->lane) {
        DeviceScope scope(l.device);
        for (int b = 0; b < kRing; b++) {
            if (l.stage[b]) (void)hipFree(l.stage[b]);
            if (l.ev_in[b]) (void)hipEventDestroy(l.ev_in[b]);
            if (l.ev_cmp[b]) (void)hipEventDestroy(l.ev_cmp[b]);
            if (l.ev_out[b]) (void)hipEventDestroy(l.ev_out[b]);
        }
        if (l.ev_done) (void)hipEventDestroy(l.ev_done);
        if (l.s_in) (void)hipStreamDestroy(l.s_in);
        if (l.s_cmp) (void)hipStreamDestroy(l.s_cmp);
        if (l.s_out) (void)hipStreamDestroy(l.s_out);
    }
    if (s->ev_start && !s->lane.empty()) {
        DeviceScope scope(s->lane[0].device);
        (void)hipEventDestroy(s->ev_start);
    }
    delete s;
    return MI355NTT_OK;
}

int mi355ntt_shards_create(mi355ntt_shards** out, const mi355ntt_ctx* const* ctxs, unsigned world, unsigned max_polys_per_piece)
{
    if (!out) return MI355NTT_EINVAL;
    *out = nullptr;
    if (!ctxs || world == 0 || world > 64) return MI355NTT_EINVAL;
    for (unsigned r = 0; r < world; r++)
        if (!ctxs[r] || mi355ntt_ctx_n(ctxs[r]) != mi355ntt_ctx_n(ctxs[0]) || mi355ntt_ctx_num_primes(ctxs[r]) != mi355ntt_ctx_num_primes(ctxs[0]))
            return MI355NTT_EINVAL;
    mi355ntt_shards* s = new (std::nothrow) mi355ntt_shards;
    if (!s) return MI355NTT_ENOMEM;
    s->world = world;
    s->n = mi355ntt_ctx_n(ctxs[0]);
    s->num_primes = mi355ntt_ctx_num_primes(ctxs[0]);
    s->piece_polys = max_polys_per_piece;
    s->lane.resize(world);
    const int root = mi355ntt_ctx_device(ctxs[0]);
    int rc = MI355NTT_OK;
    for (unsigned r = 0; r < world && rc == MI355NTT_OK; r++) {
        Lane& l = s->lane[r];
        l.ctx = ctxs[r];
        l.device = mi355ntt_ctx_device(ctxs[r]);
        DeviceScope scope(l.device);
        hipError_t e = scope.err;
        if (e == hipSuccess && l.device != root) {                 // direct copies over xGMI where the devices can reach each other
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, l.device, root) == hipSuccess && can) {
                const hipError_t pe = hipDeviceEnablePeerAccess(root, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) e = pe;
                (void)hipGetLastError();
            }
        }
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&l.s_in, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&l.s_cmp, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&l.s_out, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ev_done, hipEventDisableTiming);
        for (int b = 0; b < kRing && e == hipSuccess; b++) {
            e = hipEventCreateWithFlags(&l.ev_in[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ev_cmp[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ev_out[b], hipEventDisableTiming);
            // (the root's own shard is transformed where it lies: no staging)
            if (e == hipSuccess && r != 0 && max_polys_per_piece) e = hipMalloc((void**)&l.stage[b], (size_t)max_polys_per_piece * s->n * sizeof(mi355ntt_u64));
        }
        if (e != hipSuccess) {
            mi355ntt::record_hip_error((int)e);
            rc = e == hipErrorOutOfMemory ? MI355NTT_ENOMEM : MI355NTT_EHIP;
        }
    }
    if (rc == MI355NTT_OK) {
        DeviceScope scope(root);
        const hipError_t e = hipEventCreateWithFlags(&s->ev_start, hipEventDisableTiming);
        if (e != hipSuccess) {
            mi355ntt::record_hip_error((int)e);
            rc = MI355NTT_EHIP;
        }
    }
    if (rc != MI355NTT_OK) {
        (void)mi355ntt_shards_destroy(s);
        return rc;
    }
    *out = s;
    return MI355NTT_OK;
}

unsigned mi355ntt_shards_world(const mi355ntt_shards* s) { return s ? s->world : 0; }

/* Device-resident shards (the compute-only report of SURVEY.md 8(e)): d_shard[r] on lane r's device holds the polynomials
 * mi355ntt_shard_range(num, division, r, world) names.  One launch sequence per device, all devices concurrently; `stream` (a stream
 * of lane 0's device) is the fork and join point: the launches see what was enqueued on it before the call, work enqueued on it
 * after the call sees every shard transformed. */
int mi355ntt_shards_transform(mi355ntt_shards* s, int op, mi355ntt_u64* const* d_shard, const mi355ntt_u64* const* d_bhat_shard, unsigned num,
                              unsigned division, mi355ntt_stream stream)
{
    if (!s || !d_shard || division == 0 || division > s->num_primes) return MI355NTT_EINVAL;
    if (op != MI355NTT_OP_FORWARD && op != MI355NTT_OP_INVERSE && op != MI355NTT_OP_FORWARD_INVERSE && op != MI355NTT_OP_POLYMUL) return MI355NTT_EINVAL;
    if (op == MI355NTT_OP_POLYMUL && !d_bhat_shard) return MI355NTT_EINVAL;
    // every argument is checked BEFORE the fork: nothing is in flight when an argument error is reported
    for (unsigned r = 0; r < s->world; r++) {
        unsigned first = 0, count = 0;
        RC_TRY(mi355ntt_shard_range(num, division, r, s->world, &first, &count));
        if (count && (!d_shard[r] || (op == MI355NTT_OP_POLYMUL && !d_bhat_shard[r]))) return MI355NTT_EINVAL;
    }
    const int rc = [&]() -> int {
        {
            DeviceScope scope(s->lane[0].device);
            HIP_TRY(scope.err);
            HIP_TRY(hipEventRecord(s->ev_start, (hipStream_t)stream));
        }
        for (unsigned r = 0; r < s->world; r++) {
            unsigned first = 0, count = 0;
            RC_TRY(mi355ntt_shard_range(num, division, r, s->world, &first, &count));
            Lane& l = s->lane[r];
            DeviceScope scope(l.device);
            HIP_TRY(scope.err);
            HIP_TRY(hipStreamWaitEvent(l.s_cmp, s->ev_start, 0));
            if (count) RC_TRY(run_op(l.ctx, op, d_shard[r], d_bhat_shard ? d_bhat_shard[r] : nullptr, count, division, l.s_cmp));
            HIP_TRY(hipEventRecord(l.ev_done, l.s_cmp));
        }
        DeviceScope scope(s->lane[0].device);
        HIP_TRY(scope.err);
        for (unsigned r = 0; r < s->world; r++) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, s->lane[r].ev_done, 0));
        return MI355NTT_OK;
    }();
    if (rc != MI355NTT_OK) drain_lanes(s);      // a HIP failure half-way: no lane may still be writing the caller's buffers when the error is reported
    return rc;
}

/* Root-resident batch (the end-to-end report of SURVEY.md 8(e)): d_full [num][n] lives on lane 0's device.  Lane r > 0 receives its
 * shard in pieces by peer copies (hipMemcpyPeerAsync: xGMI where peer access exists), transforms piece k while piece k + 1 arrives
 * and piece k - 1 returns into d_full (three streams and a ring of three staging buffers per device, events only, no host
 * synchronisation); lane 0 transforms its own shard where it lies.  MI355NTT_OP_FORWARD / _INVERSE / _FORWARD_INVERSE.  Same fork /
 * join contract on `stream` as above.  At 8 GPUs the root's links bound this path (256 MiB per peer each way at configs[3]). */
int mi355ntt_shards_scatter_transform_gather(mi355ntt_shards* s, int op, mi355ntt_u64* d_full, unsigned num, unsigned division, unsigned chunks,
                                             mi355ntt_stream stream)
{
    if (!s || !d_full || division == 0 || division > s->num_primes || chunks == 0) return MI355NTT_EINVAL;
    if (op != MI355NTT_OP_FORWARD && op != MI355NTT_OP_INVERSE && op != MI355NTT_OP_FORWARD_INVERSE) return MI355NTT_EINVAL;
    if (s->world > 1 && s->piece_polys < division) return MI355NTT_EINVAL;       // (created without staging)
    const int rc = [&]() -> int {
        const int root = s->lane[0].device;
        const size_t poly_bytes = (size_t)s->n * sizeof(mi355ntt_u64);
        {
            DeviceScope scope(root);
            HIP_TRY(scope.err);
            HIP_TRY(hipEventRecord(s->ev_start, (hipStream_t)stream));
        }
        std::vector<std::pair<unsigned, unsigned>> pieces;
        for (unsigned r = 0; r < s->world; r++) {
            unsigned first = 0, count = 0;
            RC_TRY(mi355ntt_shard_range(num, division, r, s->world, &first, &count));
            Lane& l = s->lane[r];
            DeviceScope scope(l.device);
            HIP_TRY(scope.err);
            if (r == 0) {                                                       // the root's shard: in place
                HIP_TRY(hipStreamWaitEvent(l.s_cmp, s->ev_start, 0));
                if (count) RC_TRY(run_op(l.ctx, op, d_full + (size_t)first * s->n, nullptr, count, division, l.s_cmp));
                HIP_TRY(hipEventRecord(l.ev_done, l.s_cmp));
                continue;
            }
            HIP_TRY(hipStreamWaitEvent(l.s_in, s->ev_start, 0));
            piece_plan(count, division, chunks, s->piece_polys, &pieces);
            for (size_t k = 0; k < pieces.size(); k++) {
                const int b = (int)(k % kRing);
                mi355ntt_u64* src = d_full + (size_t)(first + pieces[k].first) * s->n;
                const size_t bytes = (size_t)pieces[k].second * poly_bytes;
                HIP_TRY(hipStreamWaitEvent(l.s_in, l.ev_out[b], 0));      // the buffer's previous piece (of this call or of an earlier one) has left; a never-recorded event counts as complete
                HIP_TRY(hipMemcpyPeerAsync(l.stage[b], l.device, src, root, bytes, l.s_in));
                HIP_TRY(hipEventRecord(l.ev_in[b], l.s_in));
                HIP_TRY(hipStreamWaitEvent(l.s_cmp, l.ev_in[b], 0));
                RC_TRY(run_op(l.ctx, op, l.stage[b], nullptr, pieces[k].second, division, l.s_cmp));
                HIP_TRY(hipEventRecord(l.ev_cmp[b], l.s_cmp));
                HIP_TRY(hipStreamWaitEvent(l.s_out, l.ev_cmp[b], 0));
                HIP_TRY(hipMemcpyPeerAsync(src, root, l.stage[b], l.device, bytes, l.s_out));
                HIP_TRY(hipEventRecord(l.ev_out[b], l.s_out));
            }
            if (pieces.empty()) HIP_TRY(hipStreamWaitEvent(l.s_out, s->ev_start, 0));
            HIP_TRY(hipEventRecord(l.ev_done, l.s_out));                       // (s_out is in order: behind the last piece's return)
        }
        DeviceScope scope(root);
        HIP_TRY(scope.err);
        for (unsigned r = 0; r < s->world; r++) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, s->lane[r].ev_done, 0));
        return MI355NTT_OK;
    }();
    if (rc != MI355NTT_OK) drain_lanes(s);      // (see mi355ntt_shards_transform)
    return rc;
}

}  // extern "C"
