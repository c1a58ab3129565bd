// adfp_sort.h -- stable LSD radix sort of (key, value) int pairs on the device, 8 bits per pass, for the ordering of the sample
// points by grid cell (k_scatter_sorted).  Three kernels per pass, no atomics on global memory:
//
//   k_rs_hist      a workgroup counts the digits of its tile of 2 048 keys           -> table[digit][tile]
//   k_rs_scan      exclusive scan of every digit's row of the table + the row totals   -> with the prefix over the digits (taken
//                  by k_rs_scatter itself): where each tile's run of a digit starts
//   k_rs_scatter   a workgroup places its tile: position = table entry + rank among the tile's earlier keys of that digit
//
// A key's rank inside its tile: wave w owns elements [512 w, 512 w + 512) of the tile and walks them in 8 steps of 64; in a step
// the lanes holding the same digit find each other with 8 ballots (one per digit bit), a lane's rank in the step is the number of
// matching lanes below it, and the lowest of them adds the group's size to the wave's running count of that digit (a plain LDS
// read-modify-write: one leader per digit).  Counts of earlier waves come from a first walk over the same registers.
// (rocPRIM's device radix sort did this job in 19 launches of ~6 us: 110 us per Mapper iteration.)
#pragma once
#include "adfp_device.h"

#define ADFP_RS_TILE 2048
struct RadixArgs {
    const int* key_in; const int* val_in; int* key_out; int* val_out;
    int* table;                // [256][ntiles] tile counts, row-scanned by k_rs_scan
    int* totals;               // [256] keys per digit
    int n, ntiles, shift;
};

// lanes of the wave whose `active` digit equals mine
ADFP_DEV unsigned long long match_digit(int d, bool active) {
    unsigned long long m = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned long long bal = __ballot(active && ((d >> b) & 1));
        m &= ((d >> b) & 1) ? bal : ~bal;
    }
    return m;
}

__global__ __launch_bounds__(256) void k_rs_hist(RadixArgs a) {
    __shared__ int s_cnt[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += 256) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * ADFP_RS_TILE + w * 512;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int i = base + s * 64 + lane;
        const bool ok = i < a.n;
        const int d = ok ? (a.key_in[i] >> a.shift) & 255 : 0;
        const unsigned long long m = match_digit(d, ok);
        if (ok && (m & ((1ull << lane) - 1ull)) == 0ull) s_cnt[w][d] += __popcll(m);      // the lowest matching lane
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    const int t = threadIdx.x;
    a.table[(long long)t * a.ntiles + blockIdx.x] = (s_cnt[0][t] + s_cnt[1][t]) + (s_cnt[2][t] + s_cnt[3][t]);
}

// One workgroup per digit: exclusive scan of that digit's row of tile counts (contiguous), the row's total into totals[digit].
// The prefix over the DIGITS is left to k_rs_scatter (256 numbers, every workgroup scans them itself).  (One workgroup over the
// whole 256 x tiles table was a chain of barriers and global round trips: 19-62 us per pass in three variants.)
__global__ __launch_bounds__(256) void k_rs_scan(int* __restrict__ table, int ntiles, int* __restrict__ totals) {
    __shared__ int s_w[4];
    __shared__ int s_carry;
    int* row = table + (long long)blockIdx.x * ntiles;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < ntiles ? row[i] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if ((int)(threadIdx.x & 63) >= o) inc += t; }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
        __syncthreads();
        int excl = s_carry + inc - v;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) excl += s_w[k];
        if (i < ntiles) row[i] = excl;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = excl + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = s_carry;
}

__global__ __launch_bounds__(256) void k_rs_scatter(RadixArgs a) {
    __shared__ int s_cnt[4][256];              // digits per wave, then: what the earlier waves of the tile hold of each digit
    __shared__ int s_run[4][256];              // running count inside the wave
    __shared__ int s_base[256];                // where the tile's run of each digit starts (from the scanned table)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += 256) { (&s_cnt[0][0])[i] = 0; (&s_run[0][0])[i] = 0; }
    {   // where the tile's run of digit t starts = keys of smaller digits (exclusive scan of the 256 totals) + earlier tiles' keys of t
        __shared__ int s_tw[4];
        const int tot = a.totals[threadIdx.x];
        int inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) s_tw[w] = inc;
        __syncthreads();
        int excl = inc - tot;
        for (int k = 0; k < w; ++k) excl += s_tw[k];
        s_base[threadIdx.x] = excl + a.table[(long long)threadIdx.x * a.ntiles + blockIdx.x];
    }
    __syncthreads();
    const int base = blockIdx.x * ADFP_RS_TILE + w * 512;
    int key[8], val[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int i = base + s * 64 + lane;
        const bool ok = i < a.n;
        key[s] = ok ? a.key_in[i] : 0; val[s] = ok ? a.val_in[i] : 0;
        const int d = (key[s] >> a.shift) & 255;
        const unsigned long long m = match_digit(d, ok);
        if (ok && (m & ((1ull << lane) - 1ull)) == 0ull) s_cnt[w][d] += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    {   // exclusive sum over the waves, per digit
        const int t = threadIdx.x;
        const int c0 = s_cnt[0][t], c1 = s_cnt[1][t], c2 = s_cnt[2][t];
        s_cnt[0][t] = 0; s_cnt[1][t] = c0; s_cnt[2][t] = c0 + c1; s_cnt[3][t] = c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int i = base + s * 64 + lane;
        const bool ok = i < a.n;
        const int d = (key[s] >> a.shift) & 255;
        const unsigned long long m = match_digit(d, ok);
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        if (ok) {
            const int pos = s_base[d] + s_cnt[w][d] + s_run[w][d] + __popcll(below);
            a.key_out[pos] = key[s]; a.val_out[pos] = val[s];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (ok && below == 0ull) s_run[w][d] += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
