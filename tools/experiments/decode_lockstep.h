// ARCHIVED EXPERIMENT (round 3) -- not compiled into libadfp.so.
// k_decode_lc with the low and the colour network of a tile in LOCKSTEP: layer i of both networks in one scheduling region, so
// that one network's ReLU / split instructions can fill the other network's MFMA gaps.  Needs both networks' operand sets live
// (256 registers: 512-thread workgroups, 2 waves per SIMD).  Measured per 100 000-ray batch (tools/ab_stage.py, same box):
//   sequential, 768 threads (3 waves / SIMD, the product)   1.450 ms
//   sequential, 512 threads (2 waves / SIMD)                1.535 ms
//   lockstep,   512 threads                                 1.508 ms   (-1.8 % at equal occupancy, +4 % against the product)
// To rebuild: paste the two functions below into adfp_decode_h.h (mfma_chain_h_free after mfma_chain_h, decode_two_nets_h before
// DecodeLCArgs), call decode_two_nets_h<1, 4>(...) in k_decode_lc instead of the two decode_net_h calls, launch with 512 threads.

// the same chain without the closing scheduling barrier (decode_two_nets_h wants neighbouring chains in one region)
template <int NK>
ADFP_DEV void mfma_chain_h_free(f32x16& acc, const unsigned* __restrict__ w, const f16x8* __restrict__ xh, const f16x8* __restrict__ xl) {
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512));
        const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(w + ks * 512 + 256));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[ks], acc, 0, 0, 0);
    }
}


// The two networks of a tile in LOCKSTEP (experiment, -DADFP_LC_LOCKSTEP, 512-thread workgroups = 2 waves per SIMD, 256 registers):
// layer i of the low network and layer i of the colour network sit in ONE scheduling region, so that the scheduler may place one
// network's ReLU / split instructions into the other network's MFMA gaps (a wave's own fillers are the cheap ones,
// tools/micro/mfma_fill.hip).
template <int NA, int NB>
ADFP_DEV void decode_two_nets_h(const unsigned* __restrict__ la, const unsigned* __restrict__ lb, const GridDev& ga, const GridDev& gb,
                                const float pn[3], const float pf[3], int h, int lane_off, float& amax_a, float& amax_b,
                                float* __restrict__ out_a, float* __restrict__ out_b) {
    using LA = DecLayoutH<32, NA>;
    using LB = DecLayoutH<32, NB>;
    const unsigned* wa = la + lane_off; const unsigned* wb = lb + lane_off;
    const float* bha = (const float*)la + 4 * h; const float* bhb = (const float*)lb + 4 * h;
    const float* b16a = (const float*)la + 16 * h; const float* b16b = (const float*)lb + 16 * h;
    f16x8 cha[2], cla[2], chb[2], clb[2];
    {
        float c[16];
        gather16(ga, pn, h, c);
        split8(c, cha[0], cla[0], amax_a); split8(c + 8, cha[1], cla[1], amax_a);
        gather16(gb, pn, h, c);
        split8(c, chb[0], clb[0], amax_b); split8(c + 8, chb[1], clb[1], amax_b);
    }
    f16x8 eha[6], ela[6], ehb[6], elb[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        float e[8], f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 bm = *(const f32x4*)(b16a + LA::P_BM + unit_of_h(ks, 0, j) * 4);
            e[j] = adfp_sinf(fmaf(pf[2], bm.z, fmaf(pf[1], bm.y, pf[0] * bm.x)));
            const f32x4 bn = *(const f32x4*)(b16b + LB::P_BM + unit_of_h(ks, 0, j) * 4);
            f[j] = adfp_sinf(fmaf(pf[2], bn.z, fmaf(pf[1], bn.y, pf[0] * bn.x)));
        }
        split8<false>(e, eha[ks], ela[ks], amax_a);
        split8<false>(f, ehb[ks], elb[ks], amax_b);
    }
    f32x16 acca, accb;
    f16x8 hha[2], hla[2], hhb[2], hlb[2];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        bias_init(acca, bha + LA::P_BP(i), 0);
        bias_init(accb, bhb + LB::P_BP(i), 0);
        if (i == 0) { mfma_chain_h_free<6>(acca, wa + LA::P_WP(0), eha, ela); mfma_chain_h_free<6>(accb, wb + LB::P_WP(0), ehb, elb); }
        else if (i == 3) {
            mfma_chain_h_free<6>(acca, wa + LA::P_WP(3), eha, ela); mfma_chain_h_free<6>(accb, wb + LB::P_WP(3), ehb, elb);
            mfma_chain_h_free<2>(acca, wa + LA::P_WP(3) + 6 * 512, hha, hla); mfma_chain_h_free<2>(accb, wb + LB::P_WP(3) + 6 * 512, hhb, hlb);
        } else { mfma_chain_h_free<2>(acca, wa + LA::P_WP(i), hha, hla); mfma_chain_h_free<2>(accb, wb + LB::P_WP(i), hhb, hlb); }
        relu_bias(acca, bha + LA::P_BC(i), 0);
        mfma_chain_h_free<2>(acca, wa + LA::P_WC(i), cha, cla);
        relu_bias(accb, bhb + LB::P_BC(i), 0);
        mfma_chain_h_free<2>(accb, wb + LB::P_WC(i), chb, clb);
        if (i < 4) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = acca[r];
            split8(t, hha[0], hla[0], amax_a); split8(t + 8, hha[1], hla[1], amax_a);
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = accb[r];
            split8(t, hhb[0], hlb[0], amax_b); split8(t + 8, hhb[1], hlb[1], amax_b);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int o = 0; o < NA; ++o) {
        const float* wo = (const float*)la + 16 * NA * h + LA::P_WO + o * 16;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s = fmaf(acca[r], wo[r], s);
        s += __shfl_xor(s, 32);
        out_a[o] = s + ((const float*)la)[LA::P_BO + o];
    }
#pragma unroll
    for (int o = 0; o < NB; ++o) {
        const float* wo = (const float*)lb + 16 * NB * h + LB::P_WO + o * 16;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s = fmaf(accb[r], wo[r], s);
        s += __shfl_xor(s, 32);
        out_b[o] = s + ((const float*)lb)[LB::P_BO + o];
    }
}

