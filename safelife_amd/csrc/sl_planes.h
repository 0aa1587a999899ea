// sl_planes.h -- the CA step of the row-per-lane kernels in BIT-PLANE form (rows up to 28 cells).
//
// A lane holds one board row as WS = ceil(W/2) "split halves" words (word k = cell k | cell k+WS << 16, see
// sl_rowlane.hip).  The rule of advance_board.c:34-125 only looks at ten of a cell's sixteen bits, and every
// quantity it folds over the 3x3 block is a per-bit OR, a per-bit "seen in two cells" or a population count --
// all of which are a handful of whole-row logic operations once the row is stored one 32-bit word per cell BIT
// instead of one half-word per CELL.  So the step is evaluated as
//
//   1. transposition in: the row's words plus three seam words (the cells beyond both ends of both halves:
//      the torus' wrap, so the planes come out with their neighbours across the seams already in place) go
//      through a 16 x 16 bit-matrix transposition in SWAR form -- both halves of the words at once, a byte
//      permute for the 8-stage and shift/bitop3 pairs for the 4-, 2- and 1-stages.  Plane j = bit j of every
//      cell: bit i of its low half = bit j of entry i's low cell, bit 16+i = entry i's high cell, where
//      entry 0 = (cell W-1, cell WS-1), entries 1..WS = the row's words, entry WS+1 = (cell WS, cell 0);
//   2. the neighbourhood: the rows above and below arrive by the kernels' vertical lane moves (DPP / bpermute),
//      the column triple is an XOR / majority (count) or OR / majority (flags) per plane, the row triple the
//      same on the plane shifted by one bit either way (x + x and x >> 1: the seam entries are the halo);
//   3. the rule itself on whole rows: count == 3, count in {3, 4}, frozen / preserving / inhibiting -> the cells
//      that die, the cells that are born, the cells that need a random draw (spawners);
//   4. merge out: cells that die are cleared and cells that are born take ALIVE + the inherited bits, in the
//      word domain (the new cells' five planes go back through the same transposition).
//
// Only what a wave needs is computed: the first pass transposes the four or five planes the rule's decision
// needs (alive, frozen, preserving, inhibiting, spawning: all in the cells' low bytes); the inheritance planes
// (destructible, exit, colours) are transposed and folded only if some lane of the wave has a birth, and the
// merge only runs if some cell of the wave changes.
//
// The code is written against a tiny set of primitives (bitop3, shifts, byte permute, vertical move, wave vote)
// so that the same text runs on the device (V = one 32-bit VGPR per lane) and, compiled with SL_PLANES_HOST_SIM,
// on the CPU with V = all 64 lanes of a wave (tests/sim/plane_sim.cpp checks it against the CPU oracle).
//
// Reference behaviour restated: advance_board.c:12-32 (neighbour folding), :45-47 (exit joins destructible),
// :94-124 (rule), :115-118 (spawned cell).
#pragma once

#include <stdint.h>

namespace sl {
namespace pl {

constexpr unsigned TA = 0xF0, TB = 0xCC, TC = 0xAA;      // truth-table columns of the bitop3 operands

#ifdef SL_PLANES_HOST_SIM
// ---- CPU model of a wave: 64 lanes per value ---------------------------------------------------------
#define SL_PL_DEV inline
struct V {
    uint32_t l[64];
};
inline V pconst(uint32_t c) {
    V r;
    for (int i = 0; i < 64; ++i) r.l[i] = c;
    return r;
}
#define SL_PL_LANEWISE2(name, expr)            \
    inline V name(const V &a, const V &b) {    \
        V r;                                   \
        for (int i = 0; i < 64; ++i) {         \
            const uint32_t x = a.l[i], y = b.l[i]; \
            r.l[i] = (expr);                   \
        }                                      \
        return r;                              \
    }
SL_PL_LANEWISE2(operator&, x &y)
SL_PL_LANEWISE2(operator|, x | y)
SL_PL_LANEWISE2(operator^, x ^ y)
SL_PL_LANEWISE2(padd, x + y)
inline V pshr(const V &a, int n) {
    V r;
    for (int i = 0; i < 64; ++i) r.l[i] = a.l[i] >> n;
    return r;
}
inline V pshl(const V &a, int n) {
    V r;
    for (int i = 0; i < 64; ++i) r.l[i] = a.l[i] << n;
    return r;
}
inline V pmul24(const V &a, uint32_t c) {
    V r;
    for (int i = 0; i < 64; ++i) r.l[i] = (a.l[i] & 0xFFFFFFu) * (c & 0xFFFFFFu);
    return r;
}
template <unsigned TT>
inline V pb3(const V &a, const V &b, const V &c) {      // v_bitop3_b32: bit i of TT = f(a, b, c) at (a b c) = i
    V r;
    for (int i = 0; i < 64; ++i) {
        uint32_t o = 0;
        for (int k = 0; k < 32; ++k) {
            const unsigned idx = (((a.l[i] >> k) & 1u) << 2) | (((b.l[i] >> k) & 1u) << 1) | ((c.l[i] >> k) & 1u);
            o |= ((TT >> idx) & 1u) << k;
        }
        r.l[i] = o;
    }
    return r;
}
inline V pperm(const V &s0, const V &s1, uint32_t sel) {   // v_perm_b32: selector 0-3 = bytes of s1, 4-7 = bytes of s0
    V r;
    for (int i = 0; i < 64; ++i) {
        const uint64_t src = ((uint64_t)s0.l[i] << 32) | s1.l[i];
        uint32_t o = 0;
        for (int k = 0; k < 4; ++k) {
            const unsigned s = (sel >> (8 * k)) & 0xFFu;
            const uint32_t byte = s <= 7 ? (uint32_t)((src >> (8 * s)) & 0xFFu) : (s >= 0x0D ? 0xFFu : 0u);
            o |= byte << (8 * k);
        }
        r.l[i] = o;
    }
    return r;
}
inline bool pany(const V &a) {
    for (int i = 0; i < 64; ++i)
        if (a.l[i]) return true;
    return false;
}
// vertical moves: the lane each lane reads from (-1: reads zero)
template <int VERT>
struct VCtx {
    int up_src[64], dn_src[64];
};
template <int VERT>
inline V pup(const VCtx<VERT> &vc, const V &a) {
    V r;
    for (int i = 0; i < 64; ++i) r.l[i] = vc.up_src[i] >= 0 ? a.l[vc.up_src[i]] : 0u;
    return r;
}
template <int VERT>
inline V pdn(const VCtx<VERT> &vc, const V &a) {
    V r;
    for (int i = 0; i < 64; ++i) r.l[i] = vc.dn_src[i] >= 0 ? a.l[vc.dn_src[i]] : 0u;
    return r;
}
#else
// ---- device: one VGPR per value -----------------------------------------------------------------------
#define SL_PL_DEV __device__ __forceinline__
typedef uint32_t V;
SL_PL_DEV V pconst(uint32_t c) {       // a constant the compiler must keep in a VGPR (an SGPR or literal operand
    V r = c;                            // doubles a VALU instruction's issue time on gfx950)
    asm volatile("" : "+v"(r));
    return r;
}
SL_PL_DEV V padd(V a, V b) { return a + b; }
SL_PL_DEV V pshr(V a, int n) { return a >> n; }
SL_PL_DEV V pshl(V a, int n) { return a << n; }
SL_PL_DEV V pmul24(V a, uint32_t c) { return __umul24(a, c); }
template <unsigned TT>
SL_PL_DEV V pb3(V a, V b, V c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT & 0xFFu);
}
SL_PL_DEV V pperm(V s0, V s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
SL_PL_DEV bool pany(V a) { return __ballot(a != 0) != 0; }
enum { PV_BPERM = 0, PV_SHIFT = 1, PV_ROTATE = 2 };      // = the V_* modes of sl_rowlane.hip
template <int VERT>
struct VCtx {
    int up, dn;       // PV_BPERM: ds_bpermute byte addresses of the lanes holding rows r-1 / r+1
};
template <int VERT>
SL_PL_DEV V pup(const VCtx<VERT> &vc, V v) {
    if (VERT == PV_SHIFT) return (V)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);     // wave_shr:1
    if (VERT == PV_ROTATE) return (V)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C, 0xF, 0xF, false);   // wave_ror:1
    return (V)__builtin_amdgcn_ds_bpermute(vc.up, (int)v);
}
template <int VERT>
SL_PL_DEV V pdn(const VCtx<VERT> &vc, V v) {
    if (VERT == PV_SHIFT) return (V)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);     // wave_shl:1
    if (VERT == PV_ROTATE) return (V)__builtin_amdgcn_update_dpp(0, (int)v, 0x134, 0xF, 0xF, false);   // wave_rol:1
    return (V)__builtin_amdgcn_ds_bpermute(vc.dn, (int)v);
}
#endif

// three-input functions used below
#define SL_PB3(expr, a, b, c) pb3<(unsigned)((expr)&0xFF)>((a), (b), (c))
#define PB_OR3(a, b, c) SL_PB3(TA | TB | TC, a, b, c)
#define PB_XOR3(a, b, c) SL_PB3(TA ^ TB ^ TC, a, b, c)
#define PB_MAJ(a, b, c) SL_PB3((TA & TB) | (TA & TC) | (TB & TC), a, b, c)
#define PB_SEL(a, b, m) SL_PB3((TA & TC) | (TB & ~TC), a, b, m)        /* (a & m) | (b & ~m) */

template <int W>
struct PG {
    static constexpr int WS = (W + 1) / 2, WH = W - WS, NE = WS + 2;
    static constexpr bool ODD = (W & 1) != 0;
    static_assert(NE <= 16 && W >= 4, "bit-plane step: rows of 4 to 28 cells");
    // the planes' bits that are cells of the row (the rest: seam copies and padding)
    static constexpr uint32_t REAL = (((1u << WS) - 1u) << 1) | (((1u << WH) - 1u) << 17);
};

struct PConsts {        // masks kept in VGPRs for the whole kernel
    V m4, m2, m1, one2, lo8;
};
SL_PL_DEV PConsts make_pconsts() {
    PConsts c;
    c.m4 = pconst(0x0F0F0F0Fu);
    c.m2 = pconst(0x33333333u);
    c.m1 = pconst(0x55555555u);
    c.one2 = pconst(0x00010001u);
    c.lo8 = pconst(0x00FF00FFu);
    return c;
}

// One stage of the transposition on the word pair (a = word i, b = word i + s), s < 8:
//   word i   keeps its columns with (col & s) == 0 and takes those of word i + s, moved up by s
//   word i+s keeps its columns with (col & s) != 0 and takes those of word i, moved down by s
SL_PL_DEV V bf_lo(const V &a, const V &b, int s, const V &m) { return PB_SEL(a, pshl(b, s), m); }
SL_PL_DEV V bf_hi(const V &a, const V &b, int s, const V &m) { return PB_SEL(pshr(a, s), b, m); }
// the same with one word of the pair known to be zero
SL_PL_DEV V bf_lo_a(const V &a, const V &m) { return a & m; }
SL_PL_DEV V bf_hi_a(const V &a, int s, const V &m) { return pshr(a, s) & m; }
SL_PL_DEV V bf_lo_b(const V &b, int s, const V &m) { return SL_PB3(TA & ~TB, pshl(b, s), m, m); }
SL_PL_DEV V bf_hi_b(const V &b, const V &m) { return SL_PB3(TA & ~TB, b, m, m); }

// Column triple of a plane: values from the rows above (u) and below (d).
template <int VERT>
struct Col {
    V u, d;
    SL_PL_DEV Col(const VCtx<VERT> &vc, const V &x) : u(pup<VERT>(vc, x)), d(pdn<VERT>(vc, x)) {}
};

// ---- the step ---------------------------------------------------------------------------------------
// b:      the row (split halves); replaced by the new row if anything in the wave changes
// realm:  PG<W>::REAL on lanes that own a row of a board that is being advanced, 0 elsewhere (halo copies,
//         idle lanes, boards that are through): only those lanes' cells die, are born or draw
// draw:   SPAWN only; V ok = draw(V elig) makes the random draws of the flagged cells (a plane, bits as above)
//         in row-major order and returns the cells whose draw succeeded
// returns whether any cell of the wave changed (wave-uniform); if not, b is untouched
template <int W, int VERT, bool SPAWN, class Draw>
SL_PL_DEV bool ca_planes(V (&b)[(W + 1) / 2], const VCtx<VERT> &vc, const V &realm, const PConsts &c, Draw &&draw) {
    using G = PG<W>;
    constexpr int WS = G::WS, NE = G::NE;
    // -- entries: seam words around the row's words
    const V e_first = G::ODD ? pperm(b[WS - 1], b[WS - 2], 0x05040302u)      // (cell W-1, cell WS-1)
                             : pperm(b[WS - 1], b[WS - 1], 0x01000302u);
    const V e_wrap = G::ODD ? pperm(b[0], b[WS - 1], 0x05040100u) : b[WS - 1];   // odd: (cell WS-1, cell 0)
    const V e_last = pperm(b[0], b[0], 0x01000302u);                          // (cell WS, cell 0)
#define SL_ENT(i) ((i) == 0 ? e_first : (i) < WS ? b[(i) - 1 < 0 ? 0 : (i) - 1] : (i) == WS ? e_wrap : e_last)
    // -- 8-stage, low bytes: word i = bits 0-7 of entries i (columns 0-7) and i + 8 (columns 8-15)
    V w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= NE) w[i] = pconst(0);
        else if (i + 8 >= NE) w[i] = SL_ENT(i) & c.lo8;
        else w[i] = pperm(SL_ENT(i + 8), SL_ENT(i), 0x06020400u);
    }
    // -- 4-stage: x[0..3] = bits 0-3, x[4..7] = bits 4-7 (entry index mod 4 = word index mod 4)
    V x[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = bf_lo(w[i], w[i + 4], 4, c.m4);
        x[i + 4] = bf_hi(w[i], w[i + 4], 4, c.m4);
    }
    // -- 2-stage and 1-stage for the decision's planes: alive 0, frozen 4, preserving 5, inhibiting 6, spawning 7
    const V y0 = bf_lo(x[0], x[2], 2, c.m2), y1 = bf_lo(x[1], x[3], 2, c.m2);              // bits 0-1
    const V y4 = bf_lo(x[4], x[6], 2, c.m2), y5 = bf_lo(x[5], x[7], 2, c.m2);              // bits 4-5
    const V y6 = bf_hi(x[4], x[6], 2, c.m2), y7 = bf_hi(x[5], x[7], 2, c.m2);              // bits 6-7
    const V A = bf_lo(y0, y1, 1, c.m1);
    const V Z = bf_lo(y4, y5, 1, c.m1), P = bf_hi(y4, y5, 1, c.m1);
    const V I = bf_lo(y6, y7, 1, c.m1);
    V S = A;
    if (SPAWN) S = bf_hi(y6, y7, 1, c.m1);

    // -- neighbourhood of the decision planes: column triple, then row triple
    const Col<VERT> cA(vc, A), cP(vc, P), cI(vc, I);
    const V s0 = PB_XOR3(cA.u, A, cA.d), s1 = PB_MAJ(cA.u, A, cA.d);         // alive cells in the column: s0 + 2 s1
    const V kP = PB_OR3(cP.u, P, cP.d), kI = PB_OR3(cI.u, I, cI.d);
    const V s0l = padd(s0, s0), s0r = pshr(s0, 1), s1l = padd(s1, s1), s1r = pshr(s1, 1);
    const V ones = PB_XOR3(s0l, s0, s0r), c1 = PB_MAJ(s0l, s0, s0r);         // count = ones + 2 (c1 + t) + 4 c2
    const V t = PB_XOR3(s1l, s1, s1r), c2 = PB_MAJ(s1l, s1, s1r);
    const V u = c1 ^ t;                                                      // bit 1 of the count
    const V hi3 = SL_PB3(TA | (TB & TC), c2, c1, t);                         // bits 2-3 both clear <=> !(c2 | c1 & t)
    const V b2x = SL_PB3(TA ^ (TB & TC), c2, c1, t);                         // count in 4..7 with bit 3 clear
    const V is3 = SL_PB3(TA & TB & ~TC, ones, u, hi3);
    const V is4 = SL_PB3(~TA & ~TB & TC, ones, u, b2x);
    const V fP = PB_OR3(padd(kP, kP), kP, pshr(kP, 1));
    const V fI = PB_OR3(padd(kI, kI), kI, pshr(kI, 1));
    // rule (advance_board.c:94-124)
    const V keep_a = SL_PB3(TA | TB | TC, Z, fP, is3) | is4;                 // an alive cell stays
    const V dead_free = SL_PB3(TA & ~TB & ~TC, realm, A, Z | fI);            // dead, neither frozen nor inhibited
    const V dies = SL_PB3(TA & TB & ~TC, realm, A, keep_a);
    V born = dead_free & is3;
    V spawned = pconst(0);
    if (SPAWN) {
        const Col<VERT> cS(vc, S);
        const V kS = PB_OR3(cS.u, S, cS.d);
        const V fS = PB_OR3(padd(kS, kS), kS, pshr(kS, 1));
        const V elig = SL_PB3(TA & ~TB & TC, dead_free, is3, fS);
        if (pany(elig)) spawned = draw(elig);
    }
    const V fresh = SPAWN ? (born | spawned) : born;
    const V gone = dies | fresh;                                             // cells whose old content goes
    if (!pany(gone)) return false;

    // -- new cells: ALIVE + what they inherit (only if the wave has any)
    V nm[WS];
#pragma unroll
    for (int k = 0; k < WS; ++k) nm[k] = pconst(0);
    if (pany(fresh)) {
        // destructible 3 from the low bytes; exit 8 and the colours 9-11 from the high bytes
        const V D = bf_hi(bf_hi(x[0], x[2], 2, c.m2), bf_hi(x[1], x[3], 2, c.m2), 1, c.m1);
        V h[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i >= NE) h[i] = pconst(0);
            else if (i + 8 >= NE) h[i] = pshr(SL_ENT(i), 8) & c.lo8;
            else h[i] = pperm(SL_ENT(i + 8), SL_ENT(i), 0x07030501u);
        }
        V g[4];                                                              // bits 8-11
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = bf_lo(h[i], h[i + 4], 4, c.m4);
        const V z0 = bf_lo(g[0], g[2], 2, c.m2), z1 = bf_lo(g[1], g[3], 2, c.m2);          // bits 8-9
        const V z2 = bf_hi(g[0], g[2], 2, c.m2), z3 = bf_hi(g[1], g[3], 2, c.m2);          // bits 10-11
        const V X = bf_lo(z0, z1, 1, c.m1), C0 = bf_hi(z0, z1, 1, c.m1);
        const V C1 = bf_lo(z2, z3, 1, c.m1), C2 = bf_hi(z2, z3, 1, c.m1);
        // flags an ALIVE cell hands on (advance_board.c:16-18,21,28-29): exit|destructible, colours
        V q[4];
        q[0] = SL_PB3((TA | TB) & TC, X, D, A);
        q[1] = C0 & A;
        q[2] = C1 & A;
        q[3] = C2 & A;
        V tw[4];                                                             // seen in at least two of the nine cells
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Col<VERT> cq(vc, q[j]);
            const V once = PB_OR3(cq.u, q[j], cq.d), twice = PB_MAJ(cq.u, q[j], cq.d);
            const V any2 = PB_OR3(padd(twice, twice), twice, pshr(twice, 1));              // some column has two
            const V col2 = PB_MAJ(padd(once, once), once, pshr(once, 1));                  // two columns have one
            tw[j] = any2 | col2;
        }
        V nvD = born & tw[0];
        V nvC[3];
        if (SPAWN) {
            // colours of SPAWNING cells go straight in (advance_board.c:19); a spawned cell is destructible (:117)
            const V sc[3] = {S & C0, S & C1, S & C2};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const Col<VERT> cs(vc, sc[j]);
                const V k = PB_OR3(cs.u, sc[j], cs.d);
                const V f = PB_OR3(padd(k, k), k, pshr(k, 1));
                nvC[j] = SL_PB3(TA & (TB | TC), fresh, tw[j + 1], f);
            }
            nvD = nvD | spawned;
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) nvC[j] = fresh & tw[j + 1];
        }
        // planes 0 (fresh), 3 (nvD), 9-11 (nvC) -> words, by the same transposition, 1-stage first
        const V a0 = bf_lo_a(fresh, c.m1), a1 = bf_hi_a(fresh, 1, c.m1);              // pair (0, 1): plane 1 empty
        const V a2 = bf_lo_b(nvD, 1, c.m1), a3 = bf_hi_b(nvD, c.m1);                  // pair (2, 3): plane 2 empty
        const V a8 = bf_lo_b(nvC[0], 1, c.m1), a9 = bf_hi_b(nvC[0], c.m1);            // pair (8, 9): plane 8 empty
        const V a10 = bf_lo(nvC[1], nvC[2], 1, c.m1), a11 = bf_hi(nvC[1], nvC[2], 1, c.m1);
        const V b0 = bf_lo(a0, a2, 2, c.m2), b2 = bf_hi(a0, a2, 2, c.m2);
        const V b1 = bf_lo(a1, a3, 2, c.m2), b3 = bf_hi(a1, a3, 2, c.m2);
        const V b8 = bf_lo(a8, a10, 2, c.m2), b10 = bf_hi(a8, a10, 2, c.m2);
        const V b9 = bf_lo(a9, a11, 2, c.m2), b11 = bf_hi(a9, a11, 2, c.m2);
        const V lo4[4] = {b0, b1, b2, b3}, hi4[4] = {b8, b9, b10, b11};
        V lo[8], hi[8];                                                      // low / high bytes of entries i and i + 8
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[i] = bf_lo_a(lo4[i], c.m4);
            lo[i + 4] = bf_hi_a(lo4[i], 4, c.m4);
            hi[i] = bf_lo_a(hi4[i], c.m4);
            hi[i + 4] = bf_hi_a(hi4[i], 4, c.m4);
        }
#pragma unroll
        for (int k = 0; k < WS; ++k) {
            const int i = k + 1;                                             // entry of word k
            nm[k] = i < 8 ? pperm(hi[i], lo[i], 0x06020400u) : pperm(hi[i - 8], lo[i - 8], 0x07030501u);
        }
    }
    // -- merge: a cell that dies or is born loses its old content
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        const V m = pmul24(pshr(gone, k + 1) & c.one2, 0xFFFFu);
        b[k] = SL_PB3((TA & ~TB) | TC, b[k], m, nm[k]);
    }
#undef SL_ENT
    return true;
}

}  // namespace pl
}  // namespace sl
