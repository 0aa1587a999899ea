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
inline V palign(const V &hi, const V &lo, int sh) {        // v_alignbit_b32: ({hi, lo} >> sh)[31:0]
    V r;
    for (int i = 0; i < 64; ++i) r.l[i] = (uint32_t)((((uint64_t)hi.l[i] << 32) | lo.l[i]) >> sh);
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
    int partner_src[64];        // halo lanes: the lane that owns the row they copy; every other lane: itself
};
template <int VERT>
inline V ppartner(const VCtx<VERT> &vc, const V &a) {
    V r;
    for (int i = 0; i < 64; ++i) r.l[i] = a.l[vc.partner_src[i]];
    return r;
}
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
SL_PL_DEV V palign(V hi, V lo, int sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
SL_PL_DEV bool pany(V a) { return __ballot(a != 0) != 0; }
enum { PV_BPERM = 0, PV_SHIFT = 1, PV_ROTATE = 2 };      // = the V_* modes of sl_rowlane.hip
template <int VERT>
struct VCtx {
    int up, dn;       // PV_BPERM: ds_bpermute byte addresses of the lanes holding rows r-1 / r+1
    int partner;      // PV_SHIFT, multi-step kernels: byte address of the lane that owns the row a halo lane copies
                      // (every other lane: its own)
};
template <int VERT>
SL_PL_DEV V ppartner(const VCtx<VERT> &vc, V v) {
    return VERT == PV_SHIFT ? (V)__builtin_amdgcn_ds_bpermute(vc.partner, (int)v) : v;
}
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
#define PB_SEL(a, b, m) SL_PB3((TA & TC) | (TB & ~TC), a, b, m)        /* (a & m) | (b & ~m) */

// Geometry.  Rows of up to 28 cells: ONE word per plane -- the two halves of the split layout side by side, each
// with the cells beyond its ends as seam bits (see the file header).  Rows of exactly 64 cells: TWO words per plane,
// cells 0-31 and 32-63 in order, the torus' wrap a 64-bit rotate (v_alignbit), no seam bits; the row's words go
// through two 16 x 16 transpositions (words 0-15 and 16-31) whose halves are re-paired by one byte permute each.
template <int W>
struct PG {
    static constexpr int WS = (W + 1) / 2, WH = W - WS;
    // Even rows of 30 to 60 cells (round 4; single steps only -- ca_planes): TWO words per plane, one per HALF of the
    // split layout, each laid out like a half of the one-word form -- bit 0 the cell beyond its left end, bits 1..WS
    // its cells, bit WS + 1 the cell beyond its right end -- so the neighbours are plain shifts again and the two
    // words never exchange a bit; the WS + 2 entries (seam, the row's words, seam) fill two transposition groups.
    static constexpr bool SPLIT = W > 28 && W < 64 && (W & 1) == 0 && WS + 2 <= 32;
    static constexpr int NW = (W == 64 || SPLIT) ? 2 : 1;          // words per plane
    static constexpr int NG = NW;                                   // 16-entry transposition groups
    static constexpr int NE = (NW == 1 && !SPLIT) ? WS + 2 : 16;    // entries per group (SPLIT: WS + 2 - 16 in the second)
    static constexpr bool ODD = (W & 1) != 0;
    static_assert((NW == 1 && WS + 2 <= 16 && W >= 4) || W == 64 || SPLIT, "bit-plane step: rows of 4 to 28 cells, even rows of 30 to 60, or 64");
    // the planes' bits that are cells of the row (the rest: seam copies and padding)
    static constexpr uint32_t REAL = SPLIT ? ((1u << WS) - 1u) << 1
                                           : NW == 2 ? 0xFFFFFFFFu : (((1u << WS) - 1u) << 1) | (((1u << WH) - 1u) << 17);
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

// ---- a plane: NW words, and the operations the rule needs on it ----------------------------------------------
template <int NW>
struct Pl {
    V w[NW];
};
template <unsigned TT, int NW>
SL_PL_DEV Pl<NW> qb3(const Pl<NW> &a, const Pl<NW> &b, const Pl<NW> &c) {
    Pl<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = pb3<TT>(a.w[i], b.w[i], c.w[i]);
    return r;
}
#define SL_QB3(expr, a, b, c) qb3<(unsigned)((expr)&0xFF)>((a), (b), (c))
#define QB_OR3(a, b, c) SL_QB3(TA | TB | TC, a, b, c)
#define QB_XOR3(a, b, c) SL_QB3(TA ^ TB ^ TC, a, b, c)
#define QB_MAJ(a, b, c) SL_QB3((TA & TB) | (TA & TC) | (TB & TC), a, b, c)
template <int NW>
SL_PL_DEV Pl<NW> operator&(const Pl<NW> &a, const Pl<NW> &b) {
    Pl<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = a.w[i] & b.w[i];
    return r;
}
template <int NW>
SL_PL_DEV Pl<NW> operator|(const Pl<NW> &a, const Pl<NW> &b) {
    Pl<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = a.w[i] | b.w[i];
    return r;
}
template <int NW>
SL_PL_DEV Pl<NW> operator^(const Pl<NW> &a, const Pl<NW> &b) {
    Pl<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = a.w[i] ^ b.w[i];
    return r;
}
template <int NW>
SL_PL_DEV bool qany(const Pl<NW> &a) {
    V o = a.w[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) o = o | a.w[i];
    return pany(o);
}
template <int NW>
SL_PL_DEV Pl<NW> qzero() {
    Pl<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = pconst(0);
    return r;
}
// the left / right neighbour of every cell: one word -- the seam bits are the halo, a plain shift; two words -- the
// row is a 64-bit ring
SL_PL_DEV Pl<1> qleft(const Pl<1> &a) { return Pl<1>{{padd(a.w[0], a.w[0])}}; }
SL_PL_DEV Pl<1> qright(const Pl<1> &a) { return Pl<1>{{pshr(a.w[0], 1)}}; }
SL_PL_DEV Pl<2> qleft(const Pl<2> &a) { return Pl<2>{{palign(a.w[0], a.w[1], 31), palign(a.w[1], a.w[0], 31)}}; }
SL_PL_DEV Pl<2> qright(const Pl<2> &a) { return Pl<2>{{palign(a.w[1], a.w[0], 1), palign(a.w[0], a.w[1], 1)}}; }
// (SPLIT, PG<W>::SPLIT: every word carries its own seam bits -- plain shifts, whatever the number of words)
template <bool SPLIT, int NW>
SL_PL_DEV Pl<NW> qleft_m(const Pl<NW> &a) {
    if constexpr (SPLIT) {
        Pl<NW> r;
#pragma unroll
        for (int i = 0; i < NW; ++i) r.w[i] = padd(a.w[i], a.w[i]);
        return r;
    } else {
        return qleft(a);
    }
}
template <bool SPLIT, int NW>
SL_PL_DEV Pl<NW> qright_m(const Pl<NW> &a) {
    if constexpr (SPLIT) {
        Pl<NW> r;
#pragma unroll
        for (int i = 0; i < NW; ++i) r.w[i] = pshr(a.w[i], 1);
        return r;
    } else {
        return qright(a);
    }
}
// OR / majority / parity of a plane with its two horizontal neighbours
template <int NW, bool SPLIT = false>
SL_PL_DEV Pl<NW> row_or(const Pl<NW> &a) { return QB_OR3((qleft_m<SPLIT, NW>(a)), a, (qright_m<SPLIT, NW>(a))); }

// Column triple of a plane: values from the rows above (u) and below (d).
template <int VERT, int NW>
struct Col {
    Pl<NW> u, d;
    SL_PL_DEV Col(const VCtx<VERT> &vc, const Pl<NW> &x) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            u.w[i] = pup<VERT>(vc, x.w[i]);
            d.w[i] = pdn<VERT>(vc, x.w[i]);
        }
    }
};
template <int VERT, int NW, bool SPLIT = false>
SL_PL_DEV Pl<NW> box_or(const VCtx<VERT> &vc, const Pl<NW> &x) {        // OR over the 3 x 3 block
    const Col<VERT, NW> c(vc, x);
    return row_or<NW, SPLIT>(QB_OR3(c.u, x, c.d));
}

// ---- words <-> planes ------------------------------------------------------------------------------------------------
// A transposition group: sixteen entries (split-halves words) -> the planes' 16 + 16 bits.  What the first pass leaves
// behind for the second (destructible comes out of the low bytes' 4-stage words).
struct Grp {
    V x[4];
};
// planes 0 (alive), 4 (frozen), 5 (preserving), 6 (inhibiting), 7 (spawning) of one group; ent(i) = entry i
template <int NE, bool SPAWN, class Ent>
SL_PL_DEV void group_fast(Ent &&ent, const PConsts &c, Grp &g, V &A, V &Z, V &P, V &I, V &S) {
    V w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {       // 8-stage, low bytes: word i = bits 0-7 of entries i (columns 0-7) and i + 8
        if (i >= NE) w[i] = pconst(0);
        else if (i + 8 >= NE) w[i] = ent(i) & c.lo8;
        else w[i] = pperm(ent(i + 8), ent(i), 0x06020400u);
    }
    V x[8];                             // 4-stage: x[0..3] = bits 0-3, x[4..7] = bits 4-7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = bf_lo(w[i], w[i + 4], 4, c.m4);
        x[i + 4] = bf_hi(w[i], w[i + 4], 4, c.m4);
        g.x[i] = x[i];
    }
    const V y0 = bf_lo(x[0], x[2], 2, c.m2), y1 = bf_lo(x[1], x[3], 2, c.m2);              // bits 0-1
    const V y4 = bf_lo(x[4], x[6], 2, c.m2), y5 = bf_lo(x[5], x[7], 2, c.m2);              // bits 4-5
    const V y6 = bf_hi(x[4], x[6], 2, c.m2), y7 = bf_hi(x[5], x[7], 2, c.m2);              // bits 6-7
    A = bf_lo(y0, y1, 1, c.m1);
    Z = bf_lo(y4, y5, 1, c.m1);
    P = bf_hi(y4, y5, 1, c.m1);
    I = bf_lo(y6, y7, 1, c.m1);
    S = SPAWN ? bf_hi(y6, y7, 1, c.m1) : A;
}
// planes 3 (destructible), 8 (exit), 9-11 (colours)
template <int NE, class Ent>
SL_PL_DEV void group_slow(Ent &&ent, const PConsts &c, const Grp &g, V &D, V &X, V &C0, V &C1, V &C2) {
    D = bf_hi(bf_hi(g.x[0], g.x[2], 2, c.m2), bf_hi(g.x[1], g.x[3], 2, c.m2), 1, c.m1);
    V h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= NE) h[i] = pconst(0);
        else if (i + 8 >= NE) h[i] = pshr(ent(i), 8) & c.lo8;
        else h[i] = pperm(ent(i + 8), ent(i), 0x07030501u);
    }
    V q[4];                                                              // bits 8-11
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = bf_lo(h[i], h[i + 4], 4, c.m4);
    const V z0 = bf_lo(q[0], q[2], 2, c.m2), z1 = bf_lo(q[1], q[3], 2, c.m2);              // bits 8-9
    const V z2 = bf_hi(q[0], q[2], 2, c.m2), z3 = bf_hi(q[1], q[3], 2, c.m2);              // bits 10-11
    X = bf_lo(z0, z1, 1, c.m1);
    C0 = bf_hi(z0, z1, 1, c.m1);
    C1 = bf_lo(z2, z3, 1, c.m1);
    C2 = bf_hi(z2, z3, 1, c.m1);
}
// planes 0 (fresh), 3, 9-11 of one group -> the entries' words (the same transposition, 1-stage first): lo[i] / hi[i]
// hold the low / high bytes of entries i and i + 8
SL_PL_DEV void group_back(const V &fresh, const V &nvD, const V (&nvC)[3], const PConsts &c, V (&lo)[8], V (&hi)[8]) {
    const V a0 = bf_lo_a(fresh, c.m1), a1 = bf_hi_a(fresh, 1, c.m1);              // pair (0, 1): plane 1 empty
    const V a2 = bf_lo_b(nvD, 1, c.m1), a3 = bf_hi_b(nvD, c.m1);                  // pair (2, 3): plane 2 empty
    const V a8 = bf_lo_b(nvC[0], 1, c.m1), a9 = bf_hi_b(nvC[0], c.m1);            // pair (8, 9): plane 8 empty
    const V a10 = bf_lo(nvC[1], nvC[2], 1, c.m1), a11 = bf_hi(nvC[1], nvC[2], 1, c.m1);
    const V lo4[4] = {bf_lo(a0, a2, 2, c.m2), bf_lo(a1, a3, 2, c.m2), bf_hi(a0, a2, 2, c.m2), bf_hi(a1, a3, 2, c.m2)};
    const V hi4[4] = {bf_lo(a8, a10, 2, c.m2), bf_lo(a9, a11, 2, c.m2), bf_hi(a8, a10, 2, c.m2), bf_hi(a9, a11, 2, c.m2)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] = bf_lo_a(lo4[i], c.m4);
        lo[i + 4] = bf_hi_a(lo4[i], 4, c.m4);
        hi[i] = bf_lo_a(hi4[i], c.m4);
        hi[i + 4] = bf_hi_a(hi4[i], 4, c.m4);
    }
}
SL_PL_DEV V group_entry(const V (&lo)[8], const V (&hi)[8], int i) {      // entry i of the group
    return i < 8 ? pperm(hi[i], lo[i], 0x06020400u) : pperm(hi[i - 8], lo[i - 8], 0x07030501u);
}

// The decision of advance_board.c:94-124 on whole rows.
template <int NW>
struct Verdict {
    Pl<NW> dies, born, dead_free, is3, fS;
};
template <int VERT, int NW, bool SPAWN, bool SPLIT = false>
SL_PL_DEV Verdict<NW> decide(const VCtx<VERT> &vc, const Pl<NW> &A, const Pl<NW> &Z, const Pl<NW> &P, const Pl<NW> &I,
                             const Pl<NW> &S, const Pl<NW> &realm) {
    const Col<VERT, NW> cA(vc, A);
    const Pl<NW> s0 = QB_XOR3(cA.u, A, cA.d), s1 = QB_MAJ(cA.u, A, cA.d);         // alive cells in the column: s0 + 2 s1
    const Pl<NW> s0l = qleft_m<SPLIT, NW>(s0), s0r = qright_m<SPLIT, NW>(s0), s1l = qleft_m<SPLIT, NW>(s1), s1r = qright_m<SPLIT, NW>(s1);
    const Pl<NW> ones = QB_XOR3(s0l, s0, s0r), c1 = QB_MAJ(s0l, s0, s0r);         // count = ones + 2 (c1 + t) + 4 c2
    const Pl<NW> t = QB_XOR3(s1l, s1, s1r), c2 = QB_MAJ(s1l, s1, s1r);
    const Pl<NW> u = c1 ^ t;                                                      // bit 1 of the count
    const Pl<NW> hi3 = SL_QB3(TA | (TB & TC), c2, c1, t);                         // bits 2-3 both clear <=> !(c2 | c1 & t)
    const Pl<NW> b2x = SL_QB3(TA ^ (TB & TC), c2, c1, t);                         // count in 4..7 with bit 3 clear
    Verdict<NW> v;
    v.is3 = SL_QB3(TA & TB & ~TC, ones, u, hi3);
    const Pl<NW> is4 = SL_QB3(~TA & ~TB & TC, ones, u, b2x);
    const Pl<NW> fP = box_or<VERT, NW, SPLIT>(vc, P), fI = box_or<VERT, NW, SPLIT>(vc, I);
    const Pl<NW> keep_a = SL_QB3(TA | TB | TC, Z, fP, v.is3) | is4;               // an alive cell stays
    v.dead_free = SL_QB3(TA & ~TB & ~TC, realm, A, Z | fI);                       // dead, neither frozen nor inhibited
    v.dies = SL_QB3(TA & TB & ~TC, realm, A, keep_a);
    v.born = v.dead_free & v.is3;
    v.fS = SPAWN ? box_or<VERT, NW, SPLIT>(vc, S) : A;
    return v;
}

// What a new cell inherits (advance_board.c:16-21,28-29): a flag an ALIVE neighbour carries, seen in at least two of
// the nine cells.
template <int VERT, int NW, bool SPLIT = false>
SL_PL_DEV Pl<NW> seen_twice(const VCtx<VERT> &vc, const Pl<NW> &q) {
    const Col<VERT, NW> cq(vc, q);
    const Pl<NW> once = QB_OR3(cq.u, q, cq.d), twice = QB_MAJ(cq.u, q, cq.d);
    return row_or<NW, SPLIT>(twice) | QB_MAJ((qleft_m<SPLIT, NW>(once)), once, (qright_m<SPLIT, NW>(once)));   // a column has two | two columns have one
}

// ---- the step ---------------------------------------------------------------------------------------
// b:      the row (split halves); replaced by the new row if anything in the wave changes
// realm:  PG<W>::REAL on lanes that own a row of a board that is being advanced, 0 elsewhere (halo copies,
//         idle lanes, boards that are through): only those lanes' cells die, are born or draw
// draw:   SPAWN only; ok = draw(elig) makes the random draws of the flagged cells -- planes as NW words each; one
//         word: bit 1+k = cell k, bit 17+k = cell WS+k; two words: cell 32 i + k at bit k of word i -- in row-major
//         order and returns the cells whose draw succeeded
// row_changed (optional): receives, per lane, the cells of its row that changed (non-zero = the row changed)
// returns whether any cell of the wave changed (wave-uniform); if not, b is untouched
template <int W, int VERT, bool SPAWN, class Draw>
SL_PL_DEV bool ca_planes(V (&b)[(W + 1) / 2], const VCtx<VERT> &vc, const V &realm_word, const PConsts &c, Draw &&draw,
                         V *row_changed = nullptr) {
    using G = PG<W>;
    constexpr int WS = G::WS, NE = G::NE, NW = G::NW;
    constexpr bool SPLIT = G::SPLIT;
    constexpr int NE1 = SPLIT ? WS + 2 - 16 : NE;       // entries of the second group
    // -- entries of the transposition groups
    V e_first = b[0], e_wrap = b[0], e_last = b[0];
    if (NW == 1 || SPLIT) { // seam words around the row's words
        e_first = G::ODD ? pperm(b[WS - 1], b[WS - 2], 0x05040302u)              // (cell W-1, cell WS-1)
                         : pperm(b[WS - 1], b[WS - 1], 0x01000302u);
        e_wrap = G::ODD ? pperm(b[0], b[WS - 1], 0x05040100u) : b[WS - 1];       // odd: (cell WS-1, cell 0)
        e_last = pperm(b[0], b[0], 0x01000302u);                                 // (cell WS, cell 0)
    }
    auto seamed = [&](int i) -> V {     // entry i of the seamed row: seam, words 0 .. WS-1, seam
        return i == 0 ? e_first : i < WS ? b[i - 1 < 0 ? 0 : i - 1] : i == WS ? e_wrap : i == WS + 1 ? e_last : pconst(0);
    };
    auto ent0 = [&](int i) -> V {
        if (SPLIT) return seamed(i);
        if (NW == 2) return b[i];
        return seamed(i);
    };
    auto ent1 = [&](int i) -> V {
        if (SPLIT) return seamed(16 + i);
        return b[NW == 2 ? 16 + i : 0];
    };
    // the planes of a group's two halves -> the row's plane words: one word as it is; two words: (cells 0-15 |
    // 32-47) and (16-31 | 48-63) re-paired into cells 0-31 and 32-63
    auto pair_up = [&](const V &t0, const V &t1) {
        Pl<NW> r;
        if (NW == 1) {
            r.w[0] = t0;
        } else {
            r.w[0] = pperm(t1, t0, 0x05040100u);
            r.w[NW - 1] = pperm(t1, t0, 0x07060302u);
        }
        return r;
    };
    Pl<NW> realm;
#pragma unroll
    for (int i = 0; i < NW; ++i) realm.w[i] = realm_word;
    // -- the decision's planes: alive 0, frozen 4, preserving 5, inhibiting 6, spawning 7
    Grp g0, g1;
    V a0, z0, p0, i0, s0, a1, z1, p1, i1, s1;
    group_fast<NE, SPAWN>(ent0, c, g0, a0, z0, p0, i0, s0);
    if (NW == 2) group_fast<NE1, SPAWN>(ent1, c, g1, a1, z1, p1, i1, s1);
    else a1 = a0, z1 = z0, p1 = p0, i1 = i0, s1 = s0;
    const Pl<NW> A = pair_up(a0, a1), Z = pair_up(z0, z1), P = pair_up(p0, p1), I = pair_up(i0, i1);
    const Pl<NW> S = SPAWN ? pair_up(s0, s1) : A;
    const Verdict<NW> v = decide<VERT, NW, SPAWN, SPLIT>(vc, A, Z, P, I, S, realm);
    Pl<NW> spawned = qzero<NW>();
    if (SPAWN) {
        const Pl<NW> elig = SL_QB3(TA & ~TB & TC, v.dead_free, v.is3, v.fS);
        if (qany(elig)) spawned = draw(elig);
    }
    const Pl<NW> fresh = SPAWN ? (v.born | spawned) : v.born;
    const Pl<NW> gone = v.dies | fresh;                                      // cells whose old content goes
    if (row_changed) {
        V any = gone.w[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) any = any | gone.w[i];
        *row_changed = any;
    }
    if (!qany(gone)) return false;

    // -- new cells: ALIVE + what they inherit (only if the wave has any)
    V nm[WS];
#pragma unroll
    for (int k = 0; k < WS; ++k) nm[k] = pconst(0);
    if (qany(fresh)) {
        // destructible 3 from the low bytes; exit 8 and the colours 9-11 from the high bytes
        V d0, x0, c00, c10, c20, d1, x1, c01, c11, c21;
        group_slow<NE>(ent0, c, g0, d0, x0, c00, c10, c20);
        if (NW == 2) group_slow<NE1>(ent1, c, g1, d1, x1, c01, c11, c21);
        else d1 = d0, x1 = x0, c01 = c00, c11 = c10, c21 = c20;
        const Pl<NW> D = pair_up(d0, d1), X = pair_up(x0, x1);
        const Pl<NW> C[3] = {pair_up(c00, c01), pair_up(c10, c11), pair_up(c20, c21)};
        // flags an ALIVE cell hands on: exit|destructible, colours
        const Pl<NW> twD = seen_twice<VERT, NW, SPLIT>(vc, SL_QB3((TA | TB) & TC, X, D, A));
        Pl<NW> nvD = v.born & twD;
        Pl<NW> nvC[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const Pl<NW> tw = seen_twice<VERT, NW, SPLIT>(vc, C[j] & A);
            if (SPAWN) {
                // colours of SPAWNING cells go straight in (advance_board.c:19)
                const Pl<NW> f = box_or<VERT, NW, SPLIT>(vc, S & C[j]);
                nvC[j] = SL_QB3(TA & (TB | TC), fresh, tw, f);
            } else {
                nvC[j] = fresh & tw;
            }
        }
        if (SPAWN) nvD = nvD | spawned;                                      // a spawned cell is destructible (:117)
        // planes 0 (fresh), 3 (nvD), 9-11 (nvC) -> words
#pragma unroll
        for (int grp = 0; grp < G::NG; ++grp) {
            V f, d, cc[3];
            if (NW == 1) {
                f = fresh.w[0], d = nvD.w[0];
#pragma unroll
                for (int j = 0; j < 3; ++j) cc[j] = nvC[j].w[0];
            } else {            // the group's halves back out of the plane words
                const uint32_t sel = grp == 0 ? 0x05040100u : 0x07060302u;
                f = pperm(fresh.w[NW - 1], fresh.w[0], sel);
                d = pperm(nvD.w[NW - 1], nvD.w[0], sel);
#pragma unroll
                for (int j = 0; j < 3; ++j) cc[j] = pperm(nvC[j].w[NW - 1], nvC[j].w[0], sel);
            }
            V lo[8], hi[8];
            group_back(f, d, cc, c, lo, hi);
            if (NW == 1) {
#pragma unroll
                for (int k = 0; k < WS; ++k) nm[k] = group_entry(lo, hi, k + 1);       // word k = entry k + 1
            } else if (SPLIT) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {                                          // entry 16 grp + k = word 16 grp + k - 1
                    const int word = 16 * grp + k - 1;
                    if (word >= 0 && word < WS) nm[word] = group_entry(lo, hi, k);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) nm[(16 * grp + k) % WS] = group_entry(lo, hi, k);
            }
        }
    }
    // -- merge: a cell that dies or is born loses its old content
    V gsrc[2] = {gone.w[0], gone.w[0]};
    if (NW == 2) {
        gsrc[0] = pperm(gone.w[NW - 1], gone.w[0], 0x05040100u);
        gsrc[1] = pperm(gone.w[NW - 1], gone.w[0], 0x07060302u);
    }
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        const V src = NW == 1 ? gsrc[0] : SPLIT ? gsrc[(k + 1) / 16] : gsrc[k / 16];
        const int sh = NW == 1 ? k + 1 : SPLIT ? (k + 1) % 16 : k % 16;
        const V m = pmul24(pshr(src, sh) & c.one2, 0xFFFFu);
        b[k] = SL_PB3((TA & ~TB) | TC, b[k], m, nm[k]);
    }
    return true;
}

// ---- many steps on chip: the row stays in plane form ---------------------------------------------------------
// (life_occupancy's thousand steps, advance_board n: the transposition is paid once at either end.)  Between steps
// the planes' seam bits (one-word planes) have to follow the cells they copy: instead of re-deriving them for every
// plane, the masks a step APPLIES -- the cells that go, the new cells' bits -- are completed at the seams, and
// planes with valid seams stay valid under them.  Halo LANES (the V_SHIFT layouts keep copies of a board's last and
// first row in the lanes around it) take the same masks from the lanes that own those rows.
template <int NW>
struct PState {
    Pl<NW> A, G, D, Z, P, I, S, X, C[3];       // alive 0, agent 1, destructible 3, frozen 4, preserving 5, inhibiting 6,
    Pl<NW> ever;                               // spawning 7, exit 8, colours 9-11; cells that changed at least once
};

template <int W>
SL_PL_DEV void planes_load(const V (&b)[(W + 1) / 2], const PConsts &c, PState<PG<W>::NW> &st) {
    using G = PG<W>;
    constexpr int WS = G::WS, NE = G::NE, NW = G::NW;
    constexpr bool SPLIT = G::SPLIT;
    constexpr int NE1 = SPLIT ? WS + 2 - 16 : NE;
    V e_first = b[0], e_wrap = b[0], e_last = b[0];
    if (NW == 1 || SPLIT) {
        e_first = G::ODD ? pperm(b[WS - 1], b[WS - 2], 0x05040302u) : pperm(b[WS - 1], b[WS - 1], 0x01000302u);
        e_wrap = G::ODD ? pperm(b[0], b[WS - 1], 0x05040100u) : b[WS - 1];
        e_last = pperm(b[0], b[0], 0x01000302u);
    }
    auto seamed = [&](int i) -> V {
        return i == 0 ? e_first : i < WS ? b[i - 1 < 0 ? 0 : i - 1] : i == WS ? e_wrap : i == WS + 1 ? e_last : pconst(0);
    };
    auto ent0 = [&](int i) -> V {
        if (SPLIT) return seamed(i);
        if (NW == 2) return b[i];
        return seamed(i);
    };
    auto ent1 = [&](int i) -> V {
        if (SPLIT) return seamed(16 + i);
        return b[NW == 2 ? 16 + i : 0];
    };
    V t[2][11];
#pragma unroll
    for (int g = 0; g < G::NG; ++g) {
        Grp gr;
        if (g == 0) {
            group_fast<NE, true>(ent0, c, gr, t[g][0], t[g][3], t[g][4], t[g][5], t[g][6]);
            group_slow<NE>(ent0, c, gr, t[g][2], t[g][7], t[g][8], t[g][9], t[g][10]);
        } else {
            group_fast<NE1, true>(ent1, c, gr, t[g][0], t[g][3], t[g][4], t[g][5], t[g][6]);
            group_slow<NE1>(ent1, c, gr, t[g][2], t[g][7], t[g][8], t[g][9], t[g][10]);
        }
        t[g][1] = bf_hi(bf_lo(gr.x[0], gr.x[2], 2, c.m2), bf_lo(gr.x[1], gr.x[3], 2, c.m2), 1, c.m1);     // agent: bit 1
    }
    auto put = [&](Pl<NW> &d, int j) {
        if (NW == 1) {
            d.w[0] = t[0][j];
        } else {            // (cells 0-15 | 32-47) and (16-31 | 48-63) -> cells 0-31 and 32-63
            d.w[0] = pperm(t[NW - 1][j], t[0][j], 0x05040100u);
            d.w[NW - 1] = pperm(t[NW - 1][j], t[0][j], 0x07060302u);
        }
    };
    put(st.A, 0), put(st.G, 1), put(st.D, 2), put(st.Z, 3), put(st.P, 4), put(st.I, 5), put(st.S, 6), put(st.X, 7);
    put(st.C[0], 8), put(st.C[1], 9), put(st.C[2], 10);
    st.ever = qzero<NW>();
}

// a mask over the row's cells -> the same with its seam bits filled in (one-word planes; two-word planes have none)
template <int W>
SL_PL_DEV Pl<PG<W>::NW> with_seams(const Pl<PG<W>::NW> &m) {
    using G = PG<W>;
    if constexpr (G::SPLIT) {       // every word has its own seams: bit 0 and bit WS + 1 copy cells of the other word
        constexpr int WS = G::WS;
        const V a = m.w[0], b = m.w[1];
        Pl<G::NW> out;
        out.w[0] = (a & pconst(G::REAL)) | (pshr(b, WS) & pconst(1u)) | (pshl(b, WS) & pconst(1u << (WS + 1)));   // cell W-1, cell WS
        out.w[1] = (b & pconst(G::REAL)) | (pshr(a, WS) & pconst(1u)) | (pshl(a, WS) & pconst(1u << (WS + 1)));   // cell WS-1, cell 0
        return out;
    } else if constexpr (G::NW == 2) {
        return m;
    } else {
        constexpr int WS = G::WS, WH = G::WH;
        const V x = m.w[0];
        V r = x & pconst(G::REAL);
        r = SL_PB3((TA & TB) | TC, pshr(x, 16 + WH), pconst(1u), r);                       // bit 0        <- cell W-1
        r = SL_PB3((TA & TB) | TC, pshr(x, 16 - WS), pconst(1u << (WS + 1)), r);           // bit WS+1     <- cell WS
        r = SL_PB3((TA & TB) | TC, pshl(x, 16 - WS), pconst(1u << 16), r);                 // bit 16       <- cell WS-1
        r = SL_PB3((TA & TB) | TC, pshl(x, 16 + WH), pconst(1u << (17 + WH)), r);          // bit 17+WH    <- cell 0
        Pl<G::NW> out = m;
        out.w[0] = r;
        return out;
    }
}

// One step on the planes.  Returns (wave-uniform) whether any cell of the wave changed.
template <int W, int VERT, bool SPAWN, class Draw>
SL_PL_DEV bool planes_step(PState<PG<W>::NW> &st, const VCtx<VERT> &vc, const V &realm_word, Draw &&draw) {
    constexpr int NW = PG<W>::NW;
    constexpr bool SPLIT = PG<W>::SPLIT;
    Pl<NW> realm;
#pragma unroll
    for (int i = 0; i < NW; ++i) realm.w[i] = realm_word;
    const Verdict<NW> v = decide<VERT, NW, SPAWN, SPLIT>(vc, st.A, st.Z, st.P, st.I, st.S, realm);
    Pl<NW> spawned = qzero<NW>();
    if (SPAWN) {
        const Pl<NW> elig = SL_QB3(TA & ~TB & TC, v.dead_free, v.is3, v.fS);
        if (qany(elig)) spawned = draw(elig);
    }
    Pl<NW> fresh = SPAWN ? (v.born | spawned) : v.born;
    Pl<NW> gone = v.dies | fresh;
    if (!qany(gone)) return false;
    const bool births = qany(fresh);
    Pl<NW> nvD = qzero<NW>(), nvC[3] = {qzero<NW>(), qzero<NW>(), qzero<NW>()};
    if (births) {
        nvD = v.born & seen_twice<VERT, NW, SPLIT>(vc, SL_QB3((TA | TB) & TC, st.X, st.D, st.A));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const Pl<NW> tw = seen_twice<VERT, NW, SPLIT>(vc, st.C[j] & st.A);
            if (SPAWN) nvC[j] = SL_QB3(TA & (TB | TC), fresh, tw, (box_or<VERT, NW, SPLIT>(vc, st.S & st.C[j])));
            else nvC[j] = fresh & tw;
        }
        if (SPAWN) nvD = nvD | spawned;
    }
    // the masks: seams completed, and handed to the halo lanes that copy these rows
    auto spread = [&](Pl<NW> &m) {
        m = with_seams<W>(m);
#pragma unroll
        for (int i = 0; i < NW; ++i) m.w[i] = ppartner<VERT>(vc, m.w[i]);      // (a lane's own value outside the V_SHIFT halos)
    };
    spread(gone);
    if (births) {
        spread(fresh);
        spread(nvD);
#pragma unroll
        for (int j = 0; j < 3; ++j) spread(nvC[j]);
    }
    st.A = SL_QB3((TA & ~TB) | TC, st.A, gone, fresh);
    st.D = SL_QB3((TA & ~TB) | TC, st.D, gone, nvD);
#pragma unroll
    for (int j = 0; j < 3; ++j) st.C[j] = SL_QB3((TA & ~TB) | TC, st.C[j], gone, nvC[j]);
    st.G = SL_QB3(TA & ~TB, st.G, gone, gone);
    st.Z = SL_QB3(TA & ~TB, st.Z, gone, gone);
    st.P = SL_QB3(TA & ~TB, st.P, gone, gone);
    st.I = SL_QB3(TA & ~TB, st.I, gone, gone);
    st.S = SL_QB3(TA & ~TB, st.S, gone, gone);
    st.X = SL_QB3(TA & ~TB, st.X, gone, gone);
    st.ever = st.ever | gone;
    return true;
}

// The row's words after the steps: a cell that never changed keeps its word, one that did holds exactly what its
// planes say (a cell that dies is cleared, a new cell has alive / destructible / colour bits only).
template <int W>
SL_PL_DEV void planes_store(V (&b)[(W + 1) / 2], const PConsts &c, const PState<PG<W>::NW> &st) {
    using G = PG<W>;
    constexpr int WS = G::WS, NW = G::NW;
    const Pl<NW> fa = st.A & st.ever, fd = st.D & st.ever;
    const Pl<NW> fc[3] = {st.C[0] & st.ever, st.C[1] & st.ever, st.C[2] & st.ever};
    V gsrc[2] = {st.ever.w[0], st.ever.w[0]};
    if (NW == 2) {
        gsrc[0] = pperm(st.ever.w[NW - 1], st.ever.w[0], 0x05040100u);
        gsrc[1] = pperm(st.ever.w[NW - 1], st.ever.w[0], 0x07060302u);
    }
#pragma unroll
    for (int grp = 0; grp < G::NG; ++grp) {
        V f, d, cc[3];
        if (NW == 1) {
            f = fa.w[0], d = fd.w[0];
#pragma unroll
            for (int j = 0; j < 3; ++j) cc[j] = fc[j].w[0];
        } else {
            const uint32_t sel = grp == 0 ? 0x05040100u : 0x07060302u;
            f = pperm(fa.w[NW - 1], fa.w[0], sel);
            d = pperm(fd.w[NW - 1], fd.w[0], sel);
#pragma unroll
            for (int j = 0; j < 3; ++j) cc[j] = pperm(fc[j].w[NW - 1], fc[j].w[0], sel);
        }
        V lo[8], hi[8];
        group_back(f, d, cc, c, lo, hi);
#pragma unroll
        for (int k = 0; k < (NW == 1 ? WS : 16); ++k) {
            if (G::SPLIT) {         // entry 16 grp + k = word 16 grp + k - 1
                const int word = 16 * grp + k - 1;
                if (word >= 0 && word < WS) {
                    const V m = pmul24(pshr(gsrc[grp], k) & c.one2, 0xFFFFu);
                    b[word] = SL_PB3((TA & ~TB) | TC, b[word], m, group_entry(lo, hi, k));
                }
                continue;
            }
            const int word = NW == 1 ? k : 16 * grp + k;
            const V nmw = group_entry(lo, hi, NW == 1 ? k + 1 : k);
            const V m = pmul24(pshr(gsrc[grp], NW == 1 ? k + 1 : k) & c.one2, 0xFFFFu);
            b[word % WS] = SL_PB3((TA & ~TB) | TC, b[word % WS], m, nmw);
        }
    }
}

}  // namespace pl
}  // namespace sl
