// Poseidon2-BabyBear, width 16, rate 8: permutation, row sponge, 2->1 compression and the
// duplex-sponge Fiat-Shamir state. Host and gfx950 device code share this header; all values are
// Montgomery words (bb31.h).
//
// Replaces (reference): crates/cuda-common/include/poseidon2.cuh:14-202 (constants, layers),
// crates/cuda-backend/cuda/src/merkle_tree.cu:32-51,166-176 (row sponge / compress),
// crates/cuda-backend/cuda/src/sponge.cu:24-62 and crates/stark-backend/src/transcript/
// duplex_sponge.rs:62-83 (duplex sponge).
//
// CDNA4 notes: one lane owns one 16-word state in VGPRs (64 VGPRs with the 64-bit sums of a layer); round constants are
// wave-uniform literals / scalar registers. TWO statements of the permutation live here:
//   permute_ref_t   the plain one -- canonical words, one modular reduction per addition (the host's scalar fallback, the
//                   16-lane cooperative form and the yardstick of the self-test);
//   lazy::permute   what the kernels run since round 6 (3 676 VALU instead of 6 284): linear layers summed unreduced in
//                   64-bit registers, one signed Montgomery reduction per output word, the scale of the state tracked at
//                   compile time. See the comment in front of `namespace lazy`.
#pragma once
#include "bb31.h"

namespace p2 {
using bb::Fp;

#define P2_M(x) bb::cx_to_mont(x)
static constexpr uint32_t RC_INITIAL[64] = {
    P2_M(1774958255), P2_M(1185780729), P2_M(1621102414), P2_M(1796380621), P2_M(588815102),
    P2_M(1932426223), P2_M(1925334750), P2_M(747903232),  P2_M(89648862),   P2_M(360728943),
    P2_M(977184635),  P2_M(1425273457), P2_M(256487465),  P2_M(1200041953), P2_M(572403254),
    P2_M(448208942),  P2_M(1215789478), P2_M(944884184),  P2_M(953948096),  P2_M(547326025),
    P2_M(646827752),  P2_M(889997530),  P2_M(1536873262), P2_M(86189867),   P2_M(1065944411),
    P2_M(32019634),   P2_M(333311454),  P2_M(456061748),  P2_M(1963448500), P2_M(1827584334),
    P2_M(1391160226), P2_M(1348741381), P2_M(88424255),   P2_M(104111868),  P2_M(1763866748),
    P2_M(79691676),   P2_M(1988915530), P2_M(1050669594), P2_M(359890076),  P2_M(573163527),
    P2_M(222820492),  P2_M(159256268),  P2_M(669703072),  P2_M(763177444),  P2_M(889367200),
    P2_M(256335831),  P2_M(704371273),  P2_M(25886717),   P2_M(51754520),   P2_M(1833211857),
    P2_M(454499742),  P2_M(1384520381), P2_M(777848065),  P2_M(1053320300), P2_M(1851729162),
    P2_M(344647910),  P2_M(401996362),  P2_M(1046925956), P2_M(5351995),    P2_M(1212119315),
    P2_M(754867989),  P2_M(36972490),   P2_M(751272725),  P2_M(506915399)};
static constexpr uint32_t RC_TERMINAL[64] = {
    P2_M(1922082829), P2_M(1870549801), P2_M(1502529704), P2_M(1990744480), P2_M(1700391016),
    P2_M(1702593455), P2_M(321330495),  P2_M(528965731),  P2_M(183414327),  P2_M(1886297254),
    P2_M(1178602734), P2_M(1923111974), P2_M(744004766),  P2_M(549271463),  P2_M(1781349648),
    P2_M(542259047),  P2_M(1536158148), P2_M(715456982),  P2_M(503426110),  P2_M(340311124),
    P2_M(1558555932), P2_M(1226350925), P2_M(742828095),  P2_M(1338992758), P2_M(1641600456),
    P2_M(1843351545), P2_M(301835475),  P2_M(43203215),   P2_M(386838401),  P2_M(1520185679),
    P2_M(1235297680), P2_M(904680097),  P2_M(1491801617), P2_M(1581784677), P2_M(913384905),
    P2_M(247083962),  P2_M(532844013),  P2_M(107190701),  P2_M(213827818),  P2_M(1979521776),
    P2_M(1358282574), P2_M(1681743681), P2_M(1867507480), P2_M(1530706910), P2_M(507181886),
    P2_M(695185447),  P2_M(1172395131), P2_M(1250800299), P2_M(1503161625), P2_M(817684387),
    P2_M(498481458),  P2_M(494676004),  P2_M(1404253825), P2_M(108246855),  P2_M(59414691),
    P2_M(744214112),  P2_M(890862029),  P2_M(1342765939), P2_M(1417398904), P2_M(1897591937),
    P2_M(1066647396), P2_M(1682806907), P2_M(1015795079), P2_M(1619482808)};
static constexpr uint32_t RC_INTERNAL[13] = {
    P2_M(1518359488), P2_M(1765533241), P2_M(945325693), P2_M(422793067), P2_M(311365592),
    P2_M(1311448267), P2_M(1629555936), P2_M(1009879353), P2_M(190525218), P2_M(786108885),
    P2_M(557776863),  P2_M(212616710),  P2_M(605745517)};
#undef P2_M

static constexpr int WIDTH = 16;
static constexpr int RATE = 8;
static constexpr int DIGEST = 8;

// (x + rc)^7. The sum is taken as the signed representative x + rc - P in [-P, P) (one add, rc - P
// is a literal) and the four multiplies stay in signed Montgomery form (3 instructions each, bb31.h: no
// per-multiply correction); an add and a min canonicalise the result: 15 VALU.
BB_HD Fp sbox7_add(Fp x, uint32_t rc) {
    int32_t s = (int32_t)(x.v + (rc - bb::P));
    int32_t s2 = bb::smont_mul(s, s);
    int32_t s3 = bb::smont_mul(s2, s);
    int32_t s4 = bb::smont_mul(s2, s2);
    return Fp::raw(bb::from_signed(bb::smont_mul(s4, s3)));
}

// y = M4 x,  M4 = [2 3 1 1; 1 2 3 1; 1 1 2 3; 3 1 1 2]  (Plonky3's additive form: 9 additions, 2 doublings)
BB_HD void mat4(Fp *x) {
    Fp t01 = x[0] + x[1];
    Fp t23 = x[2] + x[3];
    Fp t0123 = t01 + t23;
    Fp t01123 = t0123 + x[1];
    Fp t01233 = t0123 + x[3];
    x[3] = t01233 + x[0].dbl();
    x[1] = t01123 + x[2].dbl();
    x[0] = t01123 + t01;
    x[2] = t01233 + t23;
}
// state <- M_E * state, M_E = circ(2*M4, M4, M4, M4).
// IN  = how many leading words of the state can be non-zero (16; 8 = a sponge's first permutation, capacity still zero;
//       4 = the one-permutation hash of an extension-field row): M4 of a zero block and sums over zero blocks are skipped.
// OUT8 = only words 0..7 of the result are needed (a compression, a sponge's last permutation): the other eight are left
//       unspecified. These are the same field operations with the ones on known zeros / unused words left out -- every
//       value that is computed is the value the full layer computes.
template <int IN = 16, bool OUT8 = false>
BB_HD void external_linear(Fp *s) {
    static_assert(IN == 16 || IN == 8 || IN == 4, "IN");
#pragma unroll
    for (int b = 0; b < IN / 4; b++) mat4(s + 4 * b);
    Fp sums[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        sums[j] = IN == 16 ? (s[j] + s[4 + j]) + (s[8 + j] + s[12 + j]) : IN == 8 ? s[j] + s[4 + j] : s[j];
#pragma unroll
    for (int i = 0; i < (OUT8 ? 8 : 16); i++) s[i] = i < IN ? s[i] + sums[i & 3] : sums[i & 3];
}

// x * 2^-k for a Montgomery word: reduce (x << (32 - k)) -- a single Montgomery reduction, no
// full multiply (reference: fp.h:205-214 mul_2exp_neg_n). The compiler reduces it to 6 VALU (+3 for the addition).
template <int K>
BB_HD Fp mul_2exp_neg(Fp x) {
    return Fp::raw(bb::mont_reduce((uint64_t)x.v << (32 - K)));
}
// sum +- x * 2^-8 in 8 VALU instead of 9. P = 15 * 2^27 + 1, so 2^-K = -15 * 2^(27-K) (mod P) and, with
// x = hi * 2^K + lo,  x * 2^-K = hi - lo * 15 * 2^(27-K)  (the product is < 15 * 2^27 < P: nothing to reduce). For K = 8
// both factors fit 24 bits and `sum -+ lo * c` is ONE v_mad_i32_i24; for the other K of the diagonal the product costs
// a full-rate multiply or two shifts and a subtract, which is what the reduction form costs too (measured: DESIGN §3).
BB_HD Fp add_div256(Fp sum, Fp x) {
    const uint32_t hi = x.v >> 8, m = (x.v & 255u) * (15u << 19);
    uint32_t b = sum.v - m;  // sum - m (mod P)
    b = bb::umin32(b, b + bb::P);
    const uint32_t r = b + hi;  // < P + 2^23
    return Fp::raw(bb::umin32(r, r - bb::P));
}
BB_HD Fp sub_div256(Fp sum, Fp x) {
    const uint32_t hi = x.v >> 8, m = (x.v & 255u) * (15u << 19);
    uint32_t b = sum.v + m;  // < 2P
    b = bb::umin32(b, b - bb::P);
    const uint32_t r = b - hi;  // wraps iff negative
    return Fp::raw(bb::umin32(r, r + bb::P));
}

// state <- (J + diag(V)) * state,
// V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27]
BB_HD void internal_linear(Fp *s) {
    Fp part = s[1];
#pragma unroll
    for (int i = 2; i < 16; i++) part += s[i];
    Fp sum = part + s[0];
    s[0] = part - s[0];
    s[1] = sum + s[1];
    s[2] = sum + s[2].dbl();
    s[3] = sum + s[3].halve();
    s[4] = sum + (s[4].dbl() + s[4]);
    s[5] = sum + s[5].dbl().dbl();
    s[6] = sum - s[6].halve();
    s[7] = sum - (s[7].dbl() + s[7]);
    s[8] = sum - s[8].dbl().dbl();
    s[9] = add_div256(sum, s[9]);
    s[10] = sum + mul_2exp_neg<2>(s[10]);
    s[11] = sum + mul_2exp_neg<3>(s[11]);
    s[12] = sum + mul_2exp_neg<27>(s[12]);
    s[13] = sub_div256(sum, s[13]);
    s[14] = sum - mul_2exp_neg<4>(s[14]);
    s[15] = sum - mul_2exp_neg<27>(s[15]);
}

// IN / OUT8: see external_linear (they describe the FIRST and the LAST linear layer only).
// The permutation on canonical words, one modular reduction per addition: the plain statement of the layers. The host
// runs it (scalar fallback of the transcript); the device runs permute_lazy_t below, which must equal it word for word
// (ovm_hip_host_permute_selftest compares the two on the host, the GPU suite through the Merkle kernels).
template <int IN = 16, bool OUT8 = false>
BB_HD void permute_ref_t(Fp *s) {
    external_linear<IN, false>(s);
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7_add(s[i], RC_INITIAL[16 * r + i]);
        external_linear(s);
    }
#pragma unroll
    for (int r = 0; r < 13; r++) {
        s[0] = sbox7_add(s[0], RC_INTERNAL[r]);
        internal_linear(s);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7_add(s[i], RC_TERMINAL[16 * r + i]);
        if (r == 3) external_linear<16, OUT8>(s);
        else external_linear(s);
    }
}

// ---- the permutation with lazily reduced linear layers (round 6) ------------------------------------------------------
// What the kernels run. Measured on MI355X (tools/ubench_int64.hip): a 64-bit multiply-add (v_mad_i64_i32) and a 64-bit
// add (v_lshl_add_u64) cost ONE issue slot each, like a 32-bit multiply. So the linear layers stop reducing after every
// addition (3 VALU per modular add, ~3 900 of the 6 284 VALU of permute_ref_t): they add the signed 32-bit words into
// 64-bit sums and every output word takes ONE signed Montgomery reduction (2 VALU: lo * P^-1, then a multiply-add whose
// high word is the result).
//
// Scale bookkeeping. A word v stands for the field element X with v = X * R^e (mod P), R = 2^32; e is a compile-time
// number (an exponent mod P - 1), the same for all 16 words. The S-box (four signed Montgomery products) maps scale e
// to 7e - 6; an external layer with its reduction maps e to e - 1 (the reduction divides by R). Round constants are
// added as rc * R^e, computed at compile time (LAZY below). The initial external rounds walk
// 1 -> 0 | -6 -> -7 | -55 -> -56 | -398 -> -399 | -2799 -> -2800. An internal round is
// y_i = redc(x_i * D_i + (S mod P) * K) -- a diagonal entry times a word PLUS the sum of the state is one multiply-add in
// front of the reduction -- and the constants D_i, K hold whatever power of R the round has to apply: the first internal
// round takes the state from -2800 to 1 (Montgomery form, which the S-box fixes: the middle rounds stay there), the last
// one leaves it at the scale from which the four terminal rounds arrive at e = 1 (7 is a unit mod P - 1, so the walk
// is solved backwards at compile time). No word is ever rescaled by a product of its own: the result of the last
// external layer is the Montgomery word, up to `min(r, r + P)`. All of it is exact arithmetic on representatives of
// the same field elements: the canonical result equals permute_ref_t's.
//
// Ranges (|.| of the int32 representatives; h = 2^32; 2^31 = 1.0667 P). The bounds below assume only |K| <= P/2 of a
// constant; host_poseidon2.cpp: lazy_range_audit() re-derives them from the ACTUAL tables (words stay within .70 P) and runs
// with the host self-test:
//   smont_redc(x) = (x - m P) / h with a signed word m: |result| <= |x| / h + P/2, for |x| <= 2^62.
//   S-box: any int32 may enter. |s| <= P + 33 -> |s^2| <= .969 P, |s^3| <= .955 P, |s^4| <= .941 P, |s^7| <= .921 P.
//   external layer: |column sums| <= 4 * .921 P, |z| <= 5 * .921 P, |y| <= 7 |z| < 33 P < 2^36; reduced |a| <= P/2 + 16;
//     with the next round's constant (|c| <= P/2) the S-box input is within P + 16.
//   internal round, words 1..15 bounded by b, the S-box output by .921 P (.95 P for the inputs below): |S| <= .95 P + 15 b;
//     S mod P as a centred word Sc (|Sc| <= .534 P: q = round(S / 2^27 / 15), Sc = S - q P); T = Sc K, |K| <= P/2:
//     |T| <= .125 P h (.0334 P h in the middle rounds, where K = R mod P < 2^28); y_i = redc(x_i D_i + T [+ C]) with
//     |D_i| <= P/2 and a round constant C < P riding along in front of the next S-box:
//       first round (b = P/2 + 16, and word 0's own term x_0 K0 <= .216 P h in every sum): |y_i| <= .117 P + .216 P +
//         .125 P + P/2 = .958 P, word 0 <= .841 P;
//       middle rounds: |y_i| <= .2344 b + .0334 P + P/2: .958 P -> .758 P -> .711 P -> ... (b <= .75 P from the third on),
//         word 0 <= .2158 P + .0334 P + P/2 = .75 P;  |S| <= 15.3 P < 2^35 in the second round, <= 12.2 P afterwards;
//       last round: |y_i| <= .2344 * .75 P + .125 P + P/2 = .80 P -- the inputs of the first terminal S-boxes.
namespace lazy {
constexpr uint64_t N = bb::P - 1;           // exponents of R live mod P - 1
constexpr uint32_t R_MOD_P = 0x0ffffffeu;   // 2^32 mod P
constexpr uint64_t emod(int64_t e) { return (uint64_t)(((e % (int64_t)N) + (int64_t)N) % (int64_t)N); }
constexpr uint32_t rpow(uint64_t e) { return bb::cx_powmod(R_MOD_P, e % N); }
constexpr int32_t centre(uint32_t canon) { return canon > bb::P / 2 ? (int32_t)canon - (int32_t)bb::P : (int32_t)canon; }
constexpr uint64_t inv7() {
    for (uint64_t k = 0; k < 7; k++)
        if ((1 + k * N) % 7 == 0) return (1 + k * N) / 7;
    return 0;
}
static constexpr uint64_t INV7 = inv7();
static_assert(7 * INV7 % N == 1, "7 is a unit mod P - 1");
constexpr uint64_t before_sbox(uint64_t after) { return (after + 6) % N * INV7 % N; }  // e with 7e - 6 = after
// scale of the state the S-boxes of external round r see
static constexpr uint64_t E_INITIAL[4] = {0, emod(-7), emod(-56), emod(-399)};
static constexpr uint64_t E_ENTER = emod(-2800);                 // ... the first internal round sees
static constexpr uint64_t E_ENTER_SBOX = (7 * E_ENTER + N - 6) % N;  // ... its S-box output has
static constexpr uint64_t E_T3 = before_sbox(2), E_T2 = before_sbox(E_T3 + 1), E_T1 = before_sbox(E_T2 + 1), E_T0 = before_sbox(E_T1 + 1);
static constexpr uint64_t E_TERMINAL[4] = {E_T0, E_T1, E_T2, E_T3};
static_assert((7 * E_T3 + N - 6) % N == 2 && (7 * E_T0 + N - 6) % N == (E_T1 + 1) % N, "the terminal rounds end in Montgomery form");
static constexpr int32_t ROUND_15 = 4369;  // 4369 / 2^16 = (1 / 15)(1 - 2^-16)
#define P2_INV2K(k) bb::cx_powmod(bb::cx_powmod(2, bb::P - 2), (k))
static constexpr uint32_t DIAG_CANON[16] = {
    bb::P - 2, 1, 2, P2_INV2K(1), 3, 4, bb::P - P2_INV2K(1), bb::P - 3, bb::P - 4, P2_INV2K(8),
    P2_INV2K(2), P2_INV2K(3), P2_INV2K(27), bb::P - P2_INV2K(8), bb::P - P2_INV2K(4), bb::P - P2_INV2K(27)};
#undef P2_INV2K
// the multiplier constants of one internal round that takes words at scale e_in (word 0, after its S-box, at e_in0)
// to scale e_out: y_i R^e_out = redc(x_i * d_i R^(e_out + 1 - e_in) + ...)
struct RoundK {
    int32_t diag[16];  // word i's multiplier (word 0: (d_0 + 1) R^(e_out + 1 - e_in0) in the first round, see internal_round_first)
    int32_t k_sum;     // the multiplier of Sc
    int32_t k_word0;   // first round only: x_0's share of the sum, R^(e_out + 1 - e_in0)
};
constexpr RoundK make_round(uint64_t e_in, uint64_t e_in0, uint64_t e_out, bool first) {
    RoundK k{};
    const uint32_t f = rpow(e_out + 1 + N - e_in), f0 = rpow(e_out + 1 + N - e_in0);
    for (int i = 0; i < 16; i++) k.diag[i] = centre(bb::cx_mulmod(DIAG_CANON[i], i == 0 ? f0 : f));
    if (first) k.diag[0] = centre(bb::P - f0);  // d_0 + 1 = -1
    k.k_sum = centre(f);
    k.k_word0 = centre(f0);
    return k;
}
struct Tables {
    int32_t c_initial[64];     // added in front of the S-boxes of the initial rounds
    int32_t c_terminal[64];    // ... of the terminal rounds 1..3 (round 0's ride in the last internal round's sums)
    int32_t c_internal0;       // ... of the first internal round (word 0)
    uint32_t C_internal[13];   // the constant of internal round r >= 1 as it rides in round r - 1's sum: rc R^2 (canonical)
    uint32_t C_terminal0[16];  // the constants of terminal round 0 as they ride in the last internal round's sums
    RoundK first, middle, last;
};
// rc R^e from the Montgomery word rc_m = rc R
constexpr uint32_t scaled(uint32_t rc_m, uint64_t e) { return bb::cx_mulmod(rc_m, rpow(e + N - 1)); }
constexpr Tables make_tables() {
    Tables t{};
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < 16; i++) {
            t.c_initial[16 * r + i] = centre(scaled(RC_INITIAL[16 * r + i], E_INITIAL[r]));
            t.c_terminal[16 * r + i] = centre(scaled(RC_TERMINAL[16 * r + i], E_TERMINAL[r]));
        }
    t.c_internal0 = centre(scaled(RC_INTERNAL[0], E_ENTER));
    for (int r = 0; r < 13; r++) t.C_internal[r] = scaled(RC_INTERNAL[r], 2);
    for (int i = 0; i < 16; i++) t.C_terminal0[i] = scaled(RC_TERMINAL[i], E_T0 + 1);
    t.first = make_round(E_ENTER, E_ENTER_SBOX, 1, true);
    t.middle = make_round(1, 1, 1, false);
    t.last = make_round(1, 1, E_T0, false);
    return t;
}
static constexpr Tables LAZY = make_tables();
static_assert(LAZY.middle.k_sum == (int32_t)R_MOD_P && LAZY.middle.diag[1] == (int32_t)R_MOD_P, "a middle round multiplies by Montgomery words");

// c + a with the sign extension done by the multiply-add itself (v_mad_i64_i32 a, 1, c: ONE instruction; written as
// `c + (int64) a` the compiler extends first -- a shift and a move into a register pair -- and adds then; with an opaque
// multiplier it factors sums of such terms into 64 x 32-bit products).
BB_HD int64_t add_word(int64_t c, int32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t r;
    uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, 1, %3" : "=v"(r), "=s"(carry) : "v"(a), "v"(c));
    return r;
#else
    return c + (int64_t)a;
#endif
}
BB_HD int64_t widen(int32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t r;
    uint64_t carry;
    asm("v_mad_i64_i32 %0, %1, %2, 1, 0" : "=v"(r), "=s"(carry) : "v"(a));
    return r;
#else
    return (int64_t)a;
#endif
}
BB_HD int64_t mad(int32_t a, int32_t b, int64_t c) { return c + (int64_t)a * (int64_t)b; }
// a wave-uniform constant the compiler may not reason about: it stays ONE multiply-add operand (a literal 2^28 - 2 or a
// power of two becomes 64-bit shifts, masks and a subtract)
BB_HD int32_t opaque(int32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(k));
#endif
    return k;
}
// (a 64-bit sum the compiler may not take apart: `x D + (Sc K + 0)` would be re-associated into a second multiply-add per word)
BB_HD int64_t pin(int64_t t) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(t));
#endif
    return t;
}
// s^7 R^-6 for any int32 s, as a representative within .921 P (|s| <= P + 33) / .95 P (any s)
BB_HD int32_t sbox7(int32_t s) {
    const int32_t s2 = bb::smont_mul(s, s), s3 = bb::smont_mul(s2, s), s4 = bb::smont_mul(s2, s2);
    return bb::smont_mul(s4, s3);
}
// M4 of four 64-bit sums, reduced into four words
BB_HD void mat4_redc(const int64_t *z, int32_t *out) {
    const int64_t t01 = z[0] + z[1], t23 = z[2] + z[3], t0123 = t01 + t23, t01123 = t0123 + z[1], t01233 = t0123 + z[3];
    out[3] = bb::smont_redc(t01233 + 2 * z[0]);
    out[1] = bb::smont_redc(t01123 + 2 * z[2]);
    out[0] = bb::smont_redc(t01123 + t01);
    out[2] = bb::smont_redc(t01233 + t23);
}
// state <- M_E state / R.  out_b = M4 (x_b + X), X = the sum of the four blocks (circ(2 M4, M4, M4, M4) = M4 (I + J)).
// IN / OUT8 as in p2::external_linear.
template <int IN = 16, bool OUT8 = false>
BB_HD void external_linear(int32_t *s) {
    static_assert(IN == 16 || IN == 8 || IN == 4, "IN");
    int64_t X[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        X[j] = widen(s[j]);
#pragma unroll
        for (int b = 1; b < IN / 4; b++) X[j] = add_word(X[j], s[4 * b + j]);
    }
    int32_t zero_block[4];
    if (IN < (OUT8 ? 8 : 16)) mat4_redc(X, zero_block);  // a block whose input is zero: M4 X, the same for all of them
#pragma unroll
    for (int b = 0; b < (OUT8 ? 2 : 4); b++) {
        if (4 * b < IN) {
            int64_t z[4];
#pragma unroll
            for (int k = 0; k < 4; k++) z[k] = add_word(X[k], s[4 * b + k]);
            mat4_redc(z, s + 4 * b);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) s[4 * b + k] = zero_block[k];
        }
    }
}
// S mod P as a centred word: q = round(S / 2^27 / 15) (|S / 2^27| < 250), Sc = S - q P within .534 P
BB_HD int32_t centred_sum(int64_t S) {
    const int32_t q = ((int32_t)(S >> 27) * ROUND_15 + 32768) >> 16;
    return (int32_t)mad(q, bb::NEG_P, S);
}
// One internal round: s[0] is the S-box INPUT (its round constant already added); the words leave at the scale K is made
// for. C0: the constant that rides in word 0's sum (the next internal round's), or 0; C (16 or null): the same for all
// words (the first terminal round's constants behind the last internal round).
BB_HD void internal_round(int32_t *s, const RoundK &K, uint32_t C0, const uint32_t *C) {
    s[0] = sbox7(s[0]);
    int64_t S = widen(s[0]);
#pragma unroll
    for (int i = 1; i < 16; i++) S = add_word(S, s[i]);
    const int64_t T = pin(mad(centred_sum(S), opaque(K.k_sum), 0));
#pragma unroll
    for (int i = 0; i < 16; i++) {
        int64_t t = mad(s[i], opaque(K.diag[i]), T);
        if (C) t += (int64_t)C[i];
        else if (i == 0 && C0) t += (int64_t)C0;
        s[i] = bb::smont_redc(t);
    }
}
// The first internal round: the words arrive at scale E_ENTER, word 0 leaves its S-box at another one, so the sum of the
// state is made of two terms: x_0 K0 + (the other fifteen mod P) K1. Word 0's own d_0 x_0 = -2 x_0 and its share of the
// sum are one term, -x_0 K0.
BB_HD void internal_round_first(int32_t *s, const RoundK &K, uint32_t C0) {
    s[0] = sbox7(s[0]);
    int64_t S = widen(s[1]);
#pragma unroll
    for (int i = 2; i < 16; i++) S = add_word(S, s[i]);
    const int64_t U = pin(mad(centred_sum(S), opaque(K.k_sum), 0));
    const int64_t T = pin(mad(s[0], opaque(K.k_word0), U));
    s[0] = bb::smont_redc(mad(s[0], opaque(K.diag[0]), U) + (int64_t)C0);
#pragma unroll
    for (int i = 1; i < 16; i++) s[i] = bb::smont_redc(mad(s[i], opaque(K.diag[i]), T));
}

// where the round constants come from: the default table (literals after unrolling) ...
struct TableRC {
    BB_HD int32_t c_initial(int r, int i) const { return LAZY.c_initial[16 * r + i]; }
    BB_HD int32_t c_terminal(int r, int i) const { return LAZY.c_terminal[16 * r + i]; }
    BB_HD int32_t c_internal0() const { return LAZY.c_internal0; }
    BB_HD uint32_t C_internal(int r) const { return LAZY.C_internal[r]; }
    BB_HD uint32_t C_terminal0(int i) const { return LAZY.C_terminal0[i]; }
};
// ... or memory (`rc`: 64 initial, 64 terminal, 13 internal Montgomery words), scaled as they are read:
// rc * R^e = rc_m * R^(e - 1) = mont_mul(rc_m, R^e)
static constexpr uint32_t RPOW_INITIAL[4] = {rpow(E_INITIAL[0]), rpow(E_INITIAL[1]), rpow(E_INITIAL[2]), rpow(E_INITIAL[3])};
static constexpr uint32_t RPOW_TERMINAL[4] = {rpow(E_TERMINAL[0]), rpow(E_TERMINAL[1]), rpow(E_TERMINAL[2]), rpow(E_TERMINAL[3])};
static constexpr uint32_t RPOW_ENTER = rpow(E_ENTER), RPOW_2 = rpow(2), RPOW_T0_1 = rpow(E_T0 + 1);
struct MemoryRC {
    const uint32_t *rc;
    BB_HD static int32_t centre_rt(uint32_t v) { return v > bb::P / 2 ? (int32_t)v - (int32_t)bb::P : (int32_t)v; }
    BB_HD int32_t c_initial(int r, int i) const { return centre_rt(bb::mont_mul(rc[16 * r + i], RPOW_INITIAL[r])); }
    BB_HD int32_t c_terminal(int r, int i) const { return centre_rt(bb::mont_mul(rc[64 + 16 * r + i], RPOW_TERMINAL[r])); }
    BB_HD int32_t c_internal0() const { return centre_rt(bb::mont_mul(rc[128], RPOW_ENTER)); }
    BB_HD uint32_t C_internal(int r) const { return bb::mont_mul(rc[128 + r], RPOW_2); }
    BB_HD uint32_t C_terminal0(int i) const { return bb::mont_mul(rc[64 + i], RPOW_T0_1); }
};
template <class RC, int IN, bool OUT8>
BB_HD void permute(Fp *st, const RC &rc) {
    int32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = (int32_t)st[i].v;
    external_linear<IN, false>(s);
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7(s[i] + rc.c_initial(r, i));
        external_linear(s);
    }
    s[0] += rc.c_internal0();
    internal_round_first(s, LAZY.first, rc.C_internal(1));
#pragma unroll
    for (int r = 1; r < 12; r++) internal_round(s, LAZY.middle, rc.C_internal(r + 1), nullptr);
    uint32_t Ct[16];
#pragma unroll
    for (int i = 0; i < 16; i++) Ct[i] = rc.C_terminal0(i);
    internal_round(s, LAZY.last, 0, Ct);
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7(r == 0 ? s[i] : s[i] + rc.c_terminal(r, i));
        if (r == 3) external_linear<16, OUT8>(s);
        else external_linear(s);
    }
#pragma unroll
    for (int i = 0; i < (OUT8 ? 8 : 16); i++) st[i] = Fp::raw(bb::from_signed(s[i]));
}
}  // namespace lazy

template <int IN = 16, bool OUT8 = false>
BB_HD void permute_lazy_t(Fp *st) { lazy::permute<lazy::TableRC, IN, OUT8>(st, lazy::TableRC{}); }

template <int IN = 16, bool OUT8 = false>
BB_HD void permute_t(Fp *s) {
#if defined(__HIP_DEVICE_COMPILE__)
    permute_lazy_t<IN, OUT8>(s);
#else
    permute_ref_t<IN, OUT8>(s);
#endif
}
BB_HD void permute(Fp *s) { permute_t<16, false>(s); }

// The same layers over round constants read from memory (`rc`: 64 initial, 64 terminal, 13 internal Montgomery words).
// Not used by the prover: it lets Plonky3's published known-answer vector -- whose constants come from a seeded RNG,
// not the default table -- run through THIS file's S-box and linear layers (ovm_hip_poseidon2_permute_rc): on the
// device the lazily reduced ones the kernels run (lazy::permute with the constants scaled as they are read).
BB_HD void permute_rc(Fp *s, const uint32_t *rc) {
#if defined(__HIP_DEVICE_COMPILE__)
    lazy::permute<lazy::MemoryRC, 16, false>(s, lazy::MemoryRC{rc});
    return;
#endif
    external_linear(s);
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7_add(s[i], rc[16 * r + i]);
        external_linear(s);
    }
    for (int r = 0; r < 13; r++) {
        s[0] = sbox7_add(s[0], rc[128 + r]);
        internal_linear(s);
    }
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7_add(s[i], rc[64 + 16 * r + i]);
        external_linear(s);
    }
}

#if defined(__HIPCC__)
// ---- the permutation by 16 cooperating lanes -----------------------------------------------------
// One lane per state word (a DPP row of 16; a wave64 carries four permutations). What a single lane
// does in ~7 200 dependent-ish instructions (12 us when there is nothing else to hide it behind) takes
// ~1 100 here: the S-boxes of a full round run side by side, the linear layers are a few DPP moves. For
// the LATENCY-bound uses of the permutation -- the small top layers of a Merkle tree, where a launch
// has fewer nodes than the chip has SIMDs -- not for the bulk hashing (same arithmetic, more overhead).
// Internal-layer diagonal as Montgomery constants, one per lane.
__device__ static const uint32_t COOP_DIAG[16] = {
    bb::cx_to_mont(lazy::DIAG_CANON[0]),  bb::cx_to_mont(lazy::DIAG_CANON[1]),  bb::cx_to_mont(lazy::DIAG_CANON[2]),
    bb::cx_to_mont(lazy::DIAG_CANON[3]),  bb::cx_to_mont(lazy::DIAG_CANON[4]),  bb::cx_to_mont(lazy::DIAG_CANON[5]),
    bb::cx_to_mont(lazy::DIAG_CANON[6]),  bb::cx_to_mont(lazy::DIAG_CANON[7]),  bb::cx_to_mont(lazy::DIAG_CANON[8]),
    bb::cx_to_mont(lazy::DIAG_CANON[9]),  bb::cx_to_mont(lazy::DIAG_CANON[10]), bb::cx_to_mont(lazy::DIAG_CANON[11]),
    bb::cx_to_mont(lazy::DIAG_CANON[12]), bb::cx_to_mont(lazy::DIAG_CANON[13]), bb::cx_to_mont(lazy::DIAG_CANON[14]),
    bb::cx_to_mont(lazy::DIAG_CANON[15])};
__device__ static const uint32_t COOP_RC_INITIAL[64] = {
    RC_INITIAL[0],  RC_INITIAL[1],  RC_INITIAL[2],  RC_INITIAL[3],  RC_INITIAL[4],  RC_INITIAL[5],  RC_INITIAL[6],  RC_INITIAL[7],
    RC_INITIAL[8],  RC_INITIAL[9],  RC_INITIAL[10], RC_INITIAL[11], RC_INITIAL[12], RC_INITIAL[13], RC_INITIAL[14], RC_INITIAL[15],
    RC_INITIAL[16], RC_INITIAL[17], RC_INITIAL[18], RC_INITIAL[19], RC_INITIAL[20], RC_INITIAL[21], RC_INITIAL[22], RC_INITIAL[23],
    RC_INITIAL[24], RC_INITIAL[25], RC_INITIAL[26], RC_INITIAL[27], RC_INITIAL[28], RC_INITIAL[29], RC_INITIAL[30], RC_INITIAL[31],
    RC_INITIAL[32], RC_INITIAL[33], RC_INITIAL[34], RC_INITIAL[35], RC_INITIAL[36], RC_INITIAL[37], RC_INITIAL[38], RC_INITIAL[39],
    RC_INITIAL[40], RC_INITIAL[41], RC_INITIAL[42], RC_INITIAL[43], RC_INITIAL[44], RC_INITIAL[45], RC_INITIAL[46], RC_INITIAL[47],
    RC_INITIAL[48], RC_INITIAL[49], RC_INITIAL[50], RC_INITIAL[51], RC_INITIAL[52], RC_INITIAL[53], RC_INITIAL[54], RC_INITIAL[55],
    RC_INITIAL[56], RC_INITIAL[57], RC_INITIAL[58], RC_INITIAL[59], RC_INITIAL[60], RC_INITIAL[61], RC_INITIAL[62], RC_INITIAL[63]};
__device__ static const uint32_t COOP_RC_TERMINAL[64] = {
    RC_TERMINAL[0],  RC_TERMINAL[1],  RC_TERMINAL[2],  RC_TERMINAL[3],  RC_TERMINAL[4],  RC_TERMINAL[5],  RC_TERMINAL[6],  RC_TERMINAL[7],
    RC_TERMINAL[8],  RC_TERMINAL[9],  RC_TERMINAL[10], RC_TERMINAL[11], RC_TERMINAL[12], RC_TERMINAL[13], RC_TERMINAL[14], RC_TERMINAL[15],
    RC_TERMINAL[16], RC_TERMINAL[17], RC_TERMINAL[18], RC_TERMINAL[19], RC_TERMINAL[20], RC_TERMINAL[21], RC_TERMINAL[22], RC_TERMINAL[23],
    RC_TERMINAL[24], RC_TERMINAL[25], RC_TERMINAL[26], RC_TERMINAL[27], RC_TERMINAL[28], RC_TERMINAL[29], RC_TERMINAL[30], RC_TERMINAL[31],
    RC_TERMINAL[32], RC_TERMINAL[33], RC_TERMINAL[34], RC_TERMINAL[35], RC_TERMINAL[36], RC_TERMINAL[37], RC_TERMINAL[38], RC_TERMINAL[39],
    RC_TERMINAL[40], RC_TERMINAL[41], RC_TERMINAL[42], RC_TERMINAL[43], RC_TERMINAL[44], RC_TERMINAL[45], RC_TERMINAL[46], RC_TERMINAL[47],
    RC_TERMINAL[48], RC_TERMINAL[49], RC_TERMINAL[50], RC_TERMINAL[51], RC_TERMINAL[52], RC_TERMINAL[53], RC_TERMINAL[54], RC_TERMINAL[55],
    RC_TERMINAL[56], RC_TERMINAL[57], RC_TERMINAL[58], RC_TERMINAL[59], RC_TERMINAL[60], RC_TERMINAL[61], RC_TERMINAL[62], RC_TERMINAL[63]};

template <int CTRL>
__device__ __forceinline__ Fp dpp(Fp x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return Fp::raw((uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.v, CTRL, 0xF, 0xF, false));
#else
    return x;  // (host pass of a device-only function)
#endif
}
// lane q of every quad: 2 x_q + 3 x_{q+1} + x_{q+2} + x_{q+3} (M4 is circulant), then + the sum over the 4 quads
__device__ __forceinline__ Fp coop_external(Fp x) {
    const Fp b = dpp<0x39>(x), c = dpp<0x4E>(x), d = dpp<0x93>(x);  // quad_perm [1,2,3,0], [2,3,0,1], [3,0,1,2]
    const Fp y = ((x + b) + (c + d)) + (x + b.dbl());
    const Fp t = y + dpp<0x128>(y);          // row_ror:8
    return y + (t + dpp<0x124>(t));          // row_ror:4
}
// x = this lane's state word; lane16 = its index in the state (0..15). Returns the permuted word.
__device__ __forceinline__ Fp permute_coop(Fp x, uint32_t lane16) {
    x = coop_external(x);
#pragma unroll
    for (int r = 0; r < 4; r++) x = coop_external(sbox7_add(x, COOP_RC_INITIAL[16 * r + lane16]));
    const Fp diag = Fp::raw(COOP_DIAG[lane16]);
#pragma unroll
    for (int r = 0; r < 13; r++) {
        const Fp sb = sbox7_add(x, RC_INTERNAL[r]);
        x = lane16 == 0 ? sb : x;
        Fp t = x + dpp<0x128>(x);
        t = t + dpp<0x124>(t);
        t = t + dpp<0x122>(t);
        t = t + dpp<0x121>(t);  // the sum of all 16 words, in every lane
        x = t + x * diag;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) x = coop_external(sbox7_add(x, COOP_RC_TERMINAL[16 * r + lane16]));
    return x;
}
#endif

// The transcript's permutations run on the host, one after the other (a duplex sponge is sequential:
// ~900 per config D proof, ~20 000 when a many-AIR proof absorbs its column openings). On x86 hosts
// with AVX2 they go through an 8-lane implementation (csrc/host_poseidon2.cpp, 2.2x the scalar code);
// the dispatcher falls back to permute() elsewhere.
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(__HIPCC_RTC__) && defined(__x86_64__) && defined(OVM_HOST_FAST_PERMUTE)
void host_permute(Fp *s);
#endif
BB_HD void permute_any(Fp *s) {
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(__HIPCC_RTC__) && defined(__x86_64__) && defined(OVM_HOST_FAST_PERMUTE)
    host_permute(s);
#else
    permute(s);
#endif
}

struct Digest {
    Fp d[8];
};

BB_HD Digest compress(const Digest &l, const Digest &r) {
    Fp s[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        s[i] = l.d[i];
        s[8 + i] = r.d[i];
    }
    permute_t<16, true>(s);  // (only the first 8 words of the result are the digest)
    Digest o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.d[i] = s[i];
    return o;
}

// Overwrite-mode duplex sponge (transcript). POD so it can cross the C ABI and live on device.
struct Sponge {
    Fp state[16];
    uint32_t absorb_idx;  // 0 <= absorb_idx < 8
    uint32_t sample_idx;  // 0 <= sample_idx <= 8

    BB_HD void init() {
        for (int i = 0; i < 16; i++) state[i] = Fp::zero();
        absorb_idx = 0;
        sample_idx = 0;
    }
    BB_HD void observe(Fp x) {
        state[absorb_idx++] = x;
        if (absorb_idx == RATE) {
            permute_any(state);
            absorb_idx = 0;
            sample_idx = RATE;
        }
    }
    BB_HD Fp sample() {
        if (absorb_idx != 0 || sample_idx == 0) {
            permute_any(state);
            absorb_idx = 0;
            sample_idx = RATE;
        }
        return state[--sample_idx];
    }
    BB_HD void observe_ext(const bb::FpExt &x) {
        for (int i = 0; i < 4; i++) observe(x.c[i]);
    }
    BB_HD bb::FpExt sample_ext() {
        bb::FpExt r;
        for (int i = 0; i < 4; i++) r.c[i] = sample();
        return r;
    }
    BB_HD void observe_digest(const Digest &d) {
        for (int i = 0; i < 8; i++) observe(d.d[i]);
    }
    BB_HD uint32_t sample_bits(unsigned bits) { return sample().canonical() & ((1u << bits) - 1); }
};

// The minimum witness w of a `bits`-bit proof-of-work on s: the smallest w with (s; observe(w); sample_bits(bits)) == 0,
// or ~0. Host searches only (small `bits`; the device search is ovm_hip_sponge_grind). host_poseidon2.cpp tries eight
// candidates per step with AVX2.
#if !defined(__HIPCC_RTC__) && defined(__x86_64__) && defined(OVM_HOST_FAST_PERMUTE)
uint32_t host_grind_min_witness(const Sponge &s, unsigned bits);  // (a host function; declared in both compilation passes)
#elif !defined(__HIPCC_RTC__)
inline uint32_t host_grind_min_witness(const Sponge &s, unsigned bits) {
    for (uint32_t w = 0; w < bb::P; w++) {
        Sponge t = s;
        t.observe(Fp::from_canonical(w));
        if (t.sample_bits(bits) == 0) return w;
    }
    return ~0u;
}
#endif

}  // namespace p2
