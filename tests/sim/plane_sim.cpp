// plane_sim.cpp -- CPU model of one wavefront running the bit-plane CA step of safelife_amd/csrc/sl_planes.h.
//
// TEST INFRASTRUCTURE: compiled by tests/test_plane_sim.py with g++ (-DSL_PLANES_HOST_SIM: every value of the
// kernel text becomes the 64 lanes of a wave, the vertical lane moves become index tables that follow the DPP
// wave shift / wave rotate / ds_bpermute of the three lane layouts of sl_rowlane.hip).  The test feeds random
// boards through `steps` CA steps here and through the CPU oracle and compares boards and generator states.
//
//   plane_sim <in.bin> <out.bin>
//   in : int32 {H, W, B, steps, mode (bit 0: spawner instantiation, bit 1: the row stays in plane form)}, u16 boards[B*H*W], f32 spawn_prob[B], u64 rng[B*4]
//   out: u16 boards[B*H*W], u64 rng[B*4]
#define SL_PLANES_HOST_SIM 1
#include "../../safelife_amd/csrc/sl_planes.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace sl::pl;
typedef unsigned __int128 u128;

enum { V_BPERM = 0, V_SHIFT = 1, V_ROTATE = 2 };

struct Pcg {
    u128 state, inc;
    double next() {
        const u128 mult = ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
        state = state * mult + inc;
        const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
        const uint64_t x = hi ^ lo;
        const unsigned rot = (unsigned)(hi >> 58);
        const uint64_t o = (x >> rot) | (x << ((64u - rot) & 63u));
        return (double)(o >> 11) * (1.0 / 9007199254740992.0);
    }
};

// (the lane tables carry the layout here, so one instantiation of the kernel text serves all three)
template <int W, bool SPAWN>
static void run(int VERT, int H, int B, int steps, bool persistent, std::vector<uint16_t> &boards,
                const std::vector<float> &prob, std::vector<uint64_t> &rng) {
    constexpr int WS = (W + 1) / 2;
    const int GL = H + (VERT == V_SHIFT ? 2 : 0), G = 64 / GL;
    VCtx<0> vc;
    int lane_g[64], lane_r[64];
    bool lane_real[64], lane_in[64];
    for (int l = 0; l < 64; ++l) {
        const int g = l / GL < G ? l / GL : G - 1, j = l - g * GL;
        const bool in = l < G * GL;
        lane_in[l] = in;
        lane_g[l] = g;
        if (VERT == V_SHIFT) {
            lane_real[l] = in && j >= 1 && j <= H;
            lane_r[l] = j == 0 ? H - 1 : (j == H + 1 ? 0 : j - 1);
            vc.up_src[l] = l - 1;                       // wave_shr:1 (lane 0 reads zero)
            vc.dn_src[l] = l + 1 < 64 ? l + 1 : -1;     // wave_shl:1
            vc.partner_src[l] = !in || lane_real[l] ? l : (j == 0 ? l + H : l - H);
        } else if (VERT == V_ROTATE) {
            lane_real[l] = in;
            lane_r[l] = j;
            vc.up_src[l] = (l + 63) % 64;
            vc.dn_src[l] = (l + 1) % 64;
            vc.partner_src[l] = l;
        } else {
            lane_real[l] = in;
            lane_r[l] = in ? j : 0;
            vc.up_src[l] = in ? (lane_r[l] == 0 ? l + H - 1 : l - 1) : l;
            vc.dn_src[l] = in ? (lane_r[l] == H - 1 ? l - (H - 1) : l + 1) : l;
            vc.partner_src[l] = l;
        }
        if (!in) lane_r[l] = 0;
    }
    const PConsts cst = make_pconsts();
    for (int e0 = 0; e0 < B; e0 += G) {
        const int nbb = B - e0 < G ? B - e0 : G;
        constexpr int NW = PG<W>::NW;
        auto draw = [&](const Pl<NW> &elig) {
            Pl<NW> ok = qzero<NW>();
            for (int g = 0; g < nbb; ++g) {
                Pcg gen;
                uint64_t *st = &rng[(size_t)(e0 + g) * 4];
                gen.state = ((u128)st[0] << 64) | st[1];
                gen.inc = ((u128)st[2] << 64) | st[3];
                const double p = (double)prob[e0 + g];
                for (int l = 0; l < 64; ++l) {
                    if (!lane_in[l] || lane_g[l] != g || !lane_real[l]) continue;
                    // rows are in lane order; within a row cells in order: one word = low half (bits 1..),
                    // then high half (bits 17..); two words = word 0, then word 1, bits ascending
                    if (NW == 1) {
                        for (int part = 0; part < 2; ++part)
                            for (int i = 1; i <= WS; ++i) {
                                const uint32_t bit = 1u << (16 * part + i);
                                if (elig.w[0].l[l] & bit)
                                    if (gen.next() < p) ok.w[0].l[l] |= bit;
                            }
                    } else {
                        for (int wi = 0; wi < NW; ++wi)
                            for (int i = 0; i < 32; ++i) {
                                const uint32_t bit = 1u << i;
                                if (elig.w[wi].l[l] & bit)
                                    if (gen.next() < p) ok.w[wi].l[l] |= bit;
                            }
                    }
                }
                st[0] = (uint64_t)(gen.state >> 64);
                st[1] = (uint64_t)gen.state;
            }
            return ok;
        };
        V b[WS], realm;
        auto load_rows = [&]() {
            for (int l = 0; l < 64; ++l) {
                const bool rowl = lane_in[l] && lane_g[l] < nbb;
                realm.l[l] = rowl && lane_real[l] ? PG<W>::REAL : 0u;
                const uint16_t *row = rowl ? &boards[((size_t)(e0 + lane_g[l]) * H + lane_r[l]) * W] : nullptr;
                for (int k = 0; k < WS; ++k) {
                    const uint32_t lo = row ? row[k] : 0xBEEFu, hi = row && k + WS < W ? row[k + WS] : (row ? 0u : 0xDEADu);
                    b[k].l[l] = lo | (hi << 16);
                }
            }
        };
        auto store_rows = [&]() {
            for (int l = 0; l < 64; ++l) {
                if (!(lane_in[l] && lane_g[l] < nbb && lane_real[l])) continue;
                uint16_t *row = &boards[((size_t)(e0 + lane_g[l]) * H + lane_r[l]) * W];
                for (int k = 0; k < WS; ++k) {
                    row[k] = (uint16_t)b[k].l[l];
                    if (k + WS < W) row[k + WS] = (uint16_t)(b[k].l[l] >> 16);
                }
            }
        };
        bool kept = false;
        if (persistent) {           // the rows stay in plane form for all the steps
            load_rows();
            PState<NW> st;
            planes_load<W>(b, cst, st);
            for (int s = 0; s < steps; ++s) planes_step<W, 0, SPAWN>(st, vc, realm, draw);
            planes_store<W>(b, cst, st);
            store_rows();
            kept = true;
        }
        if (!kept) {
            for (int s = 0; s < steps; ++s) {
                load_rows();
                ca_planes<W, 0, SPAWN>(b, vc, realm, cst, draw);
                store_rows();
            }
        }
    }
}

template <int W>
static bool run_w(int H, int B, int steps, int spawn, std::vector<uint16_t> &boards, const std::vector<float> &prob,
                  std::vector<uint64_t> &rng) {
    const int vert = H == 64 ? V_ROTATE : (64 / (H + 2) == 64 / H ? V_SHIFT : V_BPERM);
    if (spawn & 1) run<W, true>(vert, H, B, steps, (spawn & 2) != 0, boards, prob, rng);
    else run<W, false>(vert, H, B, steps, (spawn & 2) != 0, boards, prob, rng);
    return true;
}

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[5];
    if (fread(hdr, 4, 5, f) != 5) return 2;
    const int H = hdr[0], W = hdr[1], B = hdr[2], steps = hdr[3];
    const int spawn = hdr[4];       // bit 0: instantiation with spawners; bit 1: planes kept across the steps
    std::vector<uint16_t> boards((size_t)B * H * W);
    std::vector<float> prob(B);
    std::vector<uint64_t> rng((size_t)B * 4);
    if (fread(boards.data(), 2, boards.size(), f) != boards.size()) return 2;
    if (fread(prob.data(), 4, prob.size(), f) != prob.size()) return 2;
    if (fread(rng.data(), 8, rng.size(), f) != rng.size()) return 2;
    fclose(f);
    bool ok = false;
#define SL_W(w) if (W == w) ok = run_w<w>(H, B, steps, spawn, boards, prob, rng);
    SL_W(4) SL_W(5) SL_W(8) SL_W(10) SL_W(12) SL_W(15) SL_W(16) SL_W(20) SL_W(24) SL_W(25) SL_W(26) SL_W(27) SL_W(28) SL_W(30) SL_W(32) SL_W(40) SL_W(48) SL_W(64)
#undef SL_W
    if (!ok) return 3;
    f = fopen(argv[2], "wb");
    if (!f) return 2;
    fwrite(boards.data(), 2, boards.size(), f);
    fwrite(rng.data(), 8, rng.size(), f);
    fclose(f);
    return 0;
}
