// adfp_sort.h -- stable LSD radix sort of (key, value) int pairs on the device, 8 bits per pass, for the ordering of the sample
// points by grid cell (k_scatter_sorted).  Three kernels per pass, no atomics on global memory:
//
//   k_rs_hist      a workgroup counts the digits of its tile of 1 024 keys           -> table[digit][tile]
//   k_rs_scan      exclusive scan of every digit's row of the table + the row totals   -> with the prefix over the digits (taken
//                  by k_rs_scatter itself): where each tile's run of a digit starts
//   k_rs_scatter   a workgroup places its tile: position = table entry + rank among the tile's earlier keys of that digit
//
// A key's rank inside its tile: wave w owns elements [256 w, 256 w + 256) of the tile and walks them in 4 steps of 64; in a step
// the lanes holding the same digit find each other with 8 ballots (one per digit bit), a lane's rank in the step is the number of
// matching lanes below it, and the lowest of them adds the group's size to the wave's running count of that digit (a plain LDS
// read-modify-write: one leader per digit).  Counts of earlier waves come from a first walk over the same registers.
// (rocPRIM's device radix sort did this job in 19 launches of ~6 us: 110 us per Mapper iteration.)
#pragma once
#include "adfp_device.h"

#ifndef ADFP_RS_TILE
#define ADFP_RS_TILE 1024
#endif
#define ADFP_RS_STEPS (ADFP_RS_TILE / 256)     // 64-key steps of a wave
struct RadixArgs {
    const int* key_in; const int* val_in; int* key_out; int* val_out;
    int* table;                // [digits][ntiles] tile counts, row-scanned by k_rs_scan
    int* totals;               // [digits] keys per digit
    int n, ntiles, shift;
    // side job of one more workgroup of k_rs_hist (the first pass of the backward's sort only): out[0] = max(parts[0 .. max_n)), the
    // fold of the per-ray cotangent maxima (adfp_backward.h: max_fold_block); NULL = none
    const float* max_parts; int max_n; float* max_out;
};

// lanes of the wave whose `active` digit equals mine
template <int BITS>
ADFP_DEV unsigned long long match_digit(int d, bool active) {
    unsigned long long m = __ballot(active);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const unsigned long long bal = __ballot(active && ((d >> b) & 1));
        m &= ((d >> b) & 1) ? bal : ~bal;
    }
    return m;
}

template <int BITS>
__global__ __launch_bounds__(256) void k_rs_hist(RadixArgs a) {
    constexpr int NB = 1 << BITS;
    if (a.max_parts && (int)blockIdx.x == a.ntiles) { max_fold_block<256>(a.max_parts, a.max_n, a.max_out); return; }
    __shared__ int s_cnt[4][NB];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * NB; i += 256) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * ADFP_RS_TILE + w * (ADFP_RS_TILE / 4);
#pragma unroll
    for (int s = 0; s < ADFP_RS_STEPS; ++s) {
        const int i = base + s * 64 + lane;
        const bool ok = i < a.n;
        const int d = ok ? (a.key_in[i] >> a.shift) & (NB - 1) : 0;
        const unsigned long long m = match_digit<BITS>(d, ok);
        if (ok && (m & ((1ull << lane) - 1ull)) == 0ull) s_cnt[w][d] += __popcll(m);      // the lowest matching lane
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NB; t += 256)
        a.table[(long long)t * a.ntiles + blockIdx.x] = (s_cnt[0][t] + s_cnt[1][t]) + (s_cnt[2][t] + s_cnt[3][t]);
}

// One WAVE per digit: exclusive scan of that digit's row of tile counts (contiguous), 64 tiles per step with the running total
// carried in a register -- no barrier; the row's total into totals[digit].  The prefix over the DIGITS is left to k_rs_scatter (every
// workgroup scans the totals itself).  (One workgroup over the whole table was a chain of barriers and global round trips: 19-62 us
// per pass in three variants.)
__global__ __launch_bounds__(256) void k_rs_scan(int* __restrict__ table, int ntiles, int* __restrict__ totals) {
    const int lane = threadIdx.x & 63;
    const int digit = blockIdx.x * 4 + (threadIdx.x >> 6);
    int* row = table + (long long)digit * ntiles;
    int carry = 0;
    for (int base = 0; base < ntiles; base += 64) {
        const int i = base + lane;
        const int v = i < ntiles ? row[i] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (i < ntiles) row[i] = carry + inc - v;
        carry += __shfl(inc, 63);
    }
    if (lane == 0) totals[digit] = carry;
}

template <int BITS>
__global__ __launch_bounds__(256) void k_rs_scatter(RadixArgs a) {
    constexpr int NB = 1 << BITS, PER = NB / 256;
    __shared__ int s_cnt[4][NB];               // digits per wave, then: what the earlier waves of the tile hold of each digit
    __shared__ int s_run[4][NB];               // running count inside the wave
    __shared__ int s_base[NB];                 // where the tile's run of each digit starts (from the scanned table)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * NB; i += 256) { (&s_cnt[0][0])[i] = 0; (&s_run[0][0])[i] = 0; }
    {   // where the tile's run of digit t starts = keys of smaller digits (exclusive scan of the NB totals) + earlier tiles' keys of t;
        // a thread takes PER consecutive digits
        __shared__ int s_tw[4];
        int tot[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { tot[k] = a.totals[threadIdx.x * PER + k]; sum += tot[k]; }
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) s_tw[w] = inc;
        __syncthreads();
        int excl = inc - sum;
        for (int k = 0; k < w; ++k) excl += s_tw[k];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int t = threadIdx.x * PER + k;
            s_base[t] = excl + a.table[(long long)t * a.ntiles + blockIdx.x];
            excl += tot[k];
        }
    }
    __syncthreads();
    const int base = blockIdx.x * ADFP_RS_TILE + w * (ADFP_RS_TILE / 4);
    int key[ADFP_RS_STEPS], val[ADFP_RS_STEPS];
#pragma unroll
    for (int s = 0; s < ADFP_RS_STEPS; ++s) {
        const int i = base + s * 64 + lane;
        const bool ok = i < a.n;
        key[s] = ok ? a.key_in[i] : 0; val[s] = ok ? a.val_in[i] : 0;
        const int d = (key[s] >> a.shift) & (NB - 1);
        const unsigned long long m = match_digit<BITS>(d, ok);
        if (ok && (m & ((1ull << lane) - 1ull)) == 0ull) s_cnt[w][d] += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NB; t += 256) {   // exclusive sum over the waves, per digit
        const int c0 = s_cnt[0][t], c1 = s_cnt[1][t], c2 = s_cnt[2][t];
        s_cnt[0][t] = 0; s_cnt[1][t] = c0; s_cnt[2][t] = c0 + c1; s_cnt[3][t] = c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < ADFP_RS_STEPS; ++s) {
        const int i = base + s * 64 + lane;
        const bool ok = i < a.n;
        const int d = (key[s] >> a.shift) & (NB - 1);
        const unsigned long long m = match_digit<BITS>(d, ok);
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

// Digit width: 8 bits.  (11-bit digits -- 2 048 counters, eleven ballots per match, a [2048][tiles] table written and read with a
// stride -- were built and measured on 21-bit keys: 12.1 + 4.8 + 14.8 us per pass against 7.2 + 4.7 + 10.1, so their two passes
// cost what three 8-bit passes do; tiles of 1 024 keys instead of 2 048 -- twice the workgroups, half the serial steps in each --
// took the three passes from 65.8 to 54.0 us, tiles of 512 to 59.7: the scan's rows grow.)
#define ADFP_RS_DIGIT_BITS 8
#define ADFP_RS_DIGITS (1 << ADFP_RS_DIGIT_BITS)
