// Routing plan of one batch through ROW-SHARDED tables (cffm_amd/dist.py ShardedStep.plan): row r of the three tables
// lives on rank r % G at local row r / G.  The reference is single-device (CFFM.py:19), so this replaces nothing in it; it is
// the device side of what the three all-to-alls of a step need to know, all of it a function of the ids alone:
//
//   local_ids [n]   the owner's local row of every slot (slot = b * F + f)
//   order     [n]   slot at sorted position q, sorted by (owner, local row), stable: slots ascend inside a run of equal ids
//   uniq      [n]   index of the DISTINCT (owner, local row) pair at sorted position q (non-decreasing)
//   pos       [n]   slot -> index of its distinct pair (which record of the answer the slot reads)
//   send_rows [n]   local row of distinct pair u (the first #distinct entries are used): the request sent to the owners
//   counts    [G]   distinct pairs per owner (int64: the split sizes of the all-to-alls)
//
// Until round 3 this was ~15 torch operations (two int64 temporaries of n elements, a stable 64-bit torch.sort, cumsum,
// index_add_, scatter).  Here: pack -> ONE rocPRIM radix sort over exactly the bits that carry (owner, local row) -> head flags
// -> rocPRIM inclusive scan -> scatter.  The key is (owner << LB | local row) << SB | slot with SB = bits of n: the slot rides in
// the low bits, which are NOT sorted on - an LSD radix sort is stable and the input is in slot order, so the slots of equal
// ids stay ascending - and is read back from the sorted key, so no value array moves through the sort.
#include "internal.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

struct PlanGeo { int sb, lb, gb; };        // bits of the slot, of a local row, of an owner

inline int bits_for(int64_t n_values) {     // bits that hold 0 .. n_values - 1 (at least 1)
    int b = 1;
    while ((1ll << b) < n_values && b < 62) ++b;
    return b;
}

__global__ __launch_bounds__(256) void plan_pack_kernel(const int32_t* __restrict__ ids, int64_t n, int world, unsigned Mmax, PlanGeo g,
                                                        unsigned long long* __restrict__ keys, int32_t* __restrict__ local_ids,
                                                        long long* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < world) counts[i] = 0;
    if (i >= n) return;
    unsigned id = (unsigned)ids[i];
    id = id < Mmax ? id : Mmax - 1;                        // an id outside [0, M) must not index beyond counts[world) / the key bits
    const unsigned owner = id % (unsigned)world, local = id / (unsigned)world;
    local_ids[i] = (int32_t)local;
    keys[i] = ((((unsigned long long)owner << g.lb) | local) << g.sb) | (unsigned long long)i;
}

// head flag of every sorted position (1 where the (owner, local row) pair differs from the one before) and the slot it holds
__global__ __launch_bounds__(256) void plan_heads_kernel(const unsigned long long* __restrict__ sorted, int64_t n, PlanGeo g,
                                                         int32_t* __restrict__ head, int32_t* __restrict__ order) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const unsigned long long k = sorted[q];
    order[q] = (int32_t)(k & ((1ull << g.sb) - 1));
    head[q] = (q == 0 || (sorted[q - 1] >> g.sb) != (k >> g.sb)) ? 1 : 0;
}

__global__ __launch_bounds__(256) void plan_scatter_kernel(const unsigned long long* __restrict__ sorted, const int32_t* __restrict__ head,
                                                           const int32_t* __restrict__ incl, const int32_t* __restrict__ order,
                                                           int64_t n, PlanGeo g, int32_t* __restrict__ uniq, int32_t* __restrict__ pos,
                                                           int32_t* __restrict__ send_rows, long long* __restrict__ counts) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const int u = incl[q] - 1;
    uniq[q] = u;
    pos[order[q]] = u;
    if (head[q]) {
        const unsigned long long comp = sorted[q] >> g.sb;
        send_rows[u] = (int32_t)(comp & ((1ull << g.lb) - 1));
        atomicAdd((unsigned long long*)&counts[comp >> g.lb], 1ull);      // integer counts: the order of the adds does not matter
    }                                                                     // send_rows beyond the distinct count is never read
}

struct Scratch { size_t keys, sorted, head, incl, tmp, tmp_bytes, total; };

inline int plan_scratch(int64_t n, Scratch* s) {
    size_t sort_tmp = 0, scan_tmp = 0;
    hipError_t e = rocprim::radix_sort_keys((void*)nullptr, sort_tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                            (size_t)n, 0u, 64u, (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    e = rocprim::inclusive_scan((void*)nullptr, scan_tmp, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, rocprim::plus<int32_t>(),
                                (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t o = 0;
    s->keys = o; o += up((size_t)n * 8);
    s->sorted = o; o += up((size_t)n * 8);
    s->head = o; o += up((size_t)n * 4);
    s->incl = o; o += up((size_t)n * 4);
    s->tmp = o;
    s->tmp_bytes = up(sort_tmp > scan_tmp ? sort_tmp : scan_tmp);
    s->total = o + s->tmp_bytes;
    return 0;
}

}  // namespace

extern "C" int64_t cffm_shard_plan_scratch_bytes(int64_t n) {
    if (n <= 0) return 256;
    Scratch s;
    if (plan_scratch(n, &s)) return -1;
    return (int64_t)s.total;
}

extern "C" int cffm_shard_plan(const int32_t* ids, int64_t n, int32_t world, int64_t M, void* scratch, int32_t* local_ids,
                               int32_t* order, int32_t* uniq, int32_t* pos, int32_t* send_rows, int64_t* counts, void* stream) {
    if (world < 1 || world > 1024 || M < 1 || n < 0 || n >= (1ll << 31)) return CFFM_ERR_BAD_SHAPE;
    if (!counts) return CFFM_ERR_BAD_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        hipError_t e = hipMemsetAsync(counts, 0, (size_t)world * 8, st);
        return e == hipSuccess ? 0 : (int)e;
    }
    if (!ids || !scratch || !local_ids || !order || !uniq || !pos || !send_rows) return CFFM_ERR_BAD_SHAPE;
    PlanGeo g;
    g.sb = bits_for(n);
    g.lb = bits_for((M + world - 1) / world);     // local rows 0 .. ceil(M / G) - 1
    g.gb = bits_for(world);
    if (g.sb + g.lb + g.gb > 64) return CFFM_ERR_BAD_SHAPE;
    Scratch s;
    int rc = plan_scratch(n, &s);
    if (rc) return rc;
    char* w = (char*)scratch;
    unsigned long long* keys = (unsigned long long*)(w + s.keys);
    unsigned long long* sorted = (unsigned long long*)(w + s.sorted);
    int32_t* head = (int32_t*)(w + s.head);
    int32_t* incl = (int32_t*)(w + s.incl);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(plan_pack_kernel, dim3(blocks), dim3(256), 0, st, ids, n, (int)world,
                       (unsigned)(M < (1ll << 31) ? M : (1ll << 31) - 1), g, keys, local_ids, (long long*)counts);
    CFFM_CHECK_LAUNCH();
    size_t tb = s.tmp_bytes;
    hipError_t e = rocprim::radix_sort_keys((void*)(w + s.tmp), tb, keys, sorted, (size_t)n, (unsigned)g.sb,
                                            (unsigned)(g.sb + g.lb + g.gb), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(plan_heads_kernel, dim3(blocks), dim3(256), 0, st, (const unsigned long long*)sorted, n, g, head, order);
    CFFM_CHECK_LAUNCH();
    tb = s.tmp_bytes;
    e = rocprim::inclusive_scan((void*)(w + s.tmp), tb, head, incl, (size_t)n, rocprim::plus<int32_t>(), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(plan_scatter_kernel, dim3(blocks), dim3(256), 0, st, (const unsigned long long*)sorted, (const int32_t*)head,
                       (const int32_t*)incl, (const int32_t*)order, n, g, uniq, pos, send_rows, (long long*)counts);
    CFFM_CHECK_LAUNCH();
    return 0;
}
