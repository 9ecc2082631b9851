// Device body of the single-workgroup LDS radix sort (shared by small_sort_kernel and the fused forward kernel).
#pragma once
#include "common.hpp"

// Single-workgroup stable LSD radix sort (4-bit digits) of up to 4096 packed keys entirely in LDS: one launch
// instead of rocPRIM's three for the 2560 lookups of a frappe batch.  Thread t owns the contiguous items
// [t*ipt, (t+1)*ipt), counts its digits in its own column of cnt[16][256] (no atomics), an exclusive scan over
// (digit-major, thread-minor) gives every (digit, thread) its first output slot, and the thread scatters its items
// in order - which is what makes the pass stable.
// The keys are either given packed (in != NULL) or packed here from the raw ids: key = (id << 32) | slot.
#define SMALL_SORT_LDS (2 * 4096 * 8 + 16 * 256 * 2 + 16)
__device__ __forceinline__ void small_sort_body(const unsigned long long* __restrict__ in, const int32_t* __restrict__ ids,
                                                unsigned long long* __restrict__ out, int n, int id_bits, char* smem) {
    unsigned long long (*buf)[4096] = reinterpret_cast<unsigned long long (*)[4096]>(smem);          // [2][4096]
    unsigned short* cnt = reinterpret_cast<unsigned short*>(smem + 2 * 4096 * 8);                   // [16 * 256]
    int* wtot = reinterpret_cast<int*>(smem + 2 * 4096 * 8 + 16 * 256 * 2);                           // [4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n; i += 256)
        buf[0][i] = in ? in[i] : (((unsigned long long)(unsigned)ids[i] << 32) | (unsigned long long)i);
    const int ipt = (n + 255) / 256;
    const int lo = min(n, tid * ipt), hi = min(n, lo + ipt);
    int cur = 0;
    __syncthreads();
    for (int shift = 32; shift < 32 + id_bits; shift += 4) {
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d * 256 + tid] = 0;
        for (int i = lo; i < hi; ++i) cnt[(int)((buf[cur][i] >> shift) & 15) * 256 + tid]++;
        __syncthreads();
        // exclusive scan of the 4096 counters in linear order; thread t handles entries [16t, 16t+16)
        int loc[16], sum = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { loc[k] = sum; sum += cnt[16 * tid + k]; }
        int incl = sum;                                            // inclusive scan across the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int base = incl - sum;
        for (int w = 0; w < wave; ++w) base += wtot[w];
#pragma unroll
        for (int k = 0; k < 16; ++k) cnt[16 * tid + k] = (unsigned short)(base + loc[k]);
        __syncthreads();
        for (int i = lo; i < hi; ++i) {
            const unsigned long long key = buf[cur][i];
            const int c = (int)((key >> shift) & 15) * 256 + tid;
            buf[cur ^ 1][cnt[c]++] = key;
        }
        cur ^= 1;
        __syncthreads();
    }
    for (int i = tid; i < n; i += 256) out[i] = buf[cur][i];
}

