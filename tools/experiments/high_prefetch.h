// Archived experiment (round 4), not compiled into the library: k_decode_high_g's tile loop as a two-deep software pipeline
// (list entry of tile t+2 and ray / depth words of tile t+1 in flight while tile t computes).  Bit-identical output, 208 VGPRs,
// 175.2 / 173.6 us per 100 000-ray batch against 177.7 / 183.9 us (A-B-A-B, profiles/r04_ab_high768.txt): inside the run-to-run
// spread, so the point loads are not what the kernel waits for.  Drop-in replacement of the loop in csrc/adfp_decode_g.h.
    // A/B: two-deep software pipeline over the tiles of a wave -- the list entry of tile t+2 and the ray / depth words of tile
    // t+1 are in flight while tile t computes (PTS_RAYS only)
    int j = threadIdx.x >> 6;
    const int sub = 16 * (g >> 1) + n;
    int tile0 = claim_tile<NT / 64>(j, &s_next, ntiles);
    int tile1 = tile0 >= 0 ? claim_tile<NT / 64>(j, &s_next, ntiles) : -1;
    int q1 = 0;
    float ro1[3] = {0.f, 0.f, 0.f}, rd1[3] = {0.f, 0.f, 0.f}; double z1 = 0.0;
    auto fetch_q = [&](int tile) { const int idx = tile * 32 + sub; return a.list[(tile >= 0 && idx < count) ? idx : 0]; };
    auto fetch_pt = [&](int q, float ro[3], float rd[3], double& z) {
        const int r = (int)((unsigned)q / (unsigned)a.P.S);
        z = a.P.z[q];
        for (int k = 0; k < 3; ++k) { ro[k] = a.P.ro[3 * r + k]; rd[k] = a.P.rd[3 * r + k]; }
    };
    int qcur = 0;
    if (tile0 >= 0) {                                    // count > 0 from here on: list[0] is a valid entry for the idle slots
        qcur = fetch_q(tile0); q1 = fetch_q(tile1);
        fetch_pt(qcur, ro1, rd1, z1);
    }
    for (int tile = tile0, tnext = tile1; tile >= 0;) {
        const int idx = tile * 32 + sub;
        const bool valid = idx < count;
        const int q = qcur;
        const int tnn = tnext >= 0 ? claim_tile<NT / 64>(j, &s_next, ntiles) : -1;
        const int q2 = fetch_q(tnn);                     // tile t+2's list entry: in flight during this tile
        double pt[3];
        for (int k = 0; k < 3; ++k) pt[k] = __dadd_rn((double)ro1[k], __dmul_rn((double)rd1[k], z1));
        qcur = q1;                                       // tile t+1's entry (loaded one tile ago) ...
        fetch_pt(qcur, ro1, rd1, z1);                    // ... and its ray / depth words: in flight during this tile
        q1 = q2;
        float pn[3], pf[2][3];
        bool pnan;
        {
            normalize3(a.nb, pt, pn);
            const float f0 = (float)pt[0], f1 = (float)pt[1], f2 = (float)pt[2];
            pnan = (pt[0] != pt[0]) | (pt[1] != pt[1]) | (pt[2] != pt[2]);
            pf[0][0] = f0; pf[0][1] = f1; pf[0][2] = f2; pf[1][0] = f0; pf[1][1] = f1; pf[1][2] = f2;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) swap_halves(pf[0][k], pf[1][k]);
        float out[2][1];
        decode_net_g<64, 1>(ldsu, a.g0, a.g1, pn, pf, lane, amax, out);
        if (valid && (g & 1) == 0) {
            const float o = g >> 1 ? out[1][0] : out[0][0];
            const float v = pnan ? __builtin_nanf("") : o;
            a.att_occ[idx] = a.single ? v : v + a.raw[4ll * q + 3];    // high + low, decoder.py:342
        }
        tile = tnext; tnext = tnn;
    }
