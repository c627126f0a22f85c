// acav_mtjump.hip -- GF(2) jump-ahead for the MT19937 stream torch.randperm draws from
// (subset_selection/code/measures/batch.py:29-32: one full randperm of the candidate list per greedy iteration).
//
// The generator is linear over GF(2): the state (19937 bits = top bit of X[m] and the 623 words after it) advances by
// one word per application of a fixed matrix F whose characteristic polynomial phi(t) has degree 19937.  For any J,
// with g(t) = t^J mod phi(t):   X[m + J + k] = XOR over { i : g_i = 1 } of X[m + i + k]   (k >= 0) -- a window of the
// stream J words ahead is a fixed XOR-combination of the 19937 windows that follow the current one.  That is what
// lets several workgroups generate ONE stream: each "lane" owns every W-th block of the stream and hops over the other
// lanes' blocks with a fixed polynomial (acav_mi.hip: k_mt_jump / k_mt_generate_lanes).
//
// Nothing here is tabulated: phi is recovered from the generator itself with Berlekamp-Massey (once per process,
// ~20 ms), t^J by square-and-multiply with the sparse reduction (phi has 135 terms).  acav_rng_jump() applies the same
// polynomials on the host -- the CPU tests pin the arithmetic against plain sequential generation.
#include <map>
#include <mutex>

#include "acav_common.h"

namespace acav {

namespace {

constexpr int MT_DEG = 19937;
constexpr int PW = (2 * MT_DEG + 63) / 64 + 1;  // 64-bit words of a product polynomial (degree < 2 * 19937)

inline uint32_t mt_next_word(const uint32_t *x)  // x -> &X[m - 624]; returns X[m]
{
    const uint32_t y = (x[0] & 0x80000000u) | (x[1] & 0x7fffffffu);
    return x[397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

struct Poly {  // bit i = coefficient of t^i
    uint64_t w[PW];
    void clear() { memset(w, 0, sizeof(w)); }
    bool bit(int i) const { return (w[i >> 6] >> (i & 63)) & 1u; }
    void flip(int i) { w[i >> 6] ^= 1ull << (i & 63); }
};

std::once_flag g_phi_once;
std::vector<int> g_phi_taps;  // exponents e < 19937 with a non-zero coefficient in phi(t) = t^19937 + sum t^e

// Berlekamp-Massey over GF(2) on one output bit of the generator (any non-trivial linear functional of the state has
// the full characteristic polynomial as its minimal polynomial: phi is primitive).
void compute_phi()
{
    const int n = 2 * MT_DEG + 64;
    std::vector<uint32_t> X(624 + (size_t)n);
    X[0] = 5489u;
    for (int i = 1; i < 624; ++i) X[i] = 1812433253u * (X[i - 1] ^ (X[i - 1] >> 30)) + (uint32_t)i;
    for (int m = 624; m < 624 + n; ++m) X[m] = mt_next_word(&X[m - 624]);
    const int NW = (MT_DEG + 1 + 63) / 64 + 1;
    std::vector<uint64_t> C(NW, 0), B(NW, 0), T(NW, 0), R(NW, 0);  // R: bit j = s[i - j]
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int i = 0; i < n; ++i) {
        const unsigned b = (X[624 + i] >> 7) & 1u;
        for (int q = NW - 1; q > 0; --q) R[q] = (R[q] << 1) | (R[q - 1] >> 63);  // R = (R << 1) | b
        R[0] = (R[0] << 1) | b;
        uint64_t acc = 0;
        for (int q = 0; q < NW; ++q) acc ^= C[q] & R[q];
        if (__builtin_parityll(acc)) {
            T = C;
            const int ws = m >> 6, bs = m & 63;  // C ^= B << m
            for (int q = NW - 1; q >= ws; --q) {
                uint64_t v = B[q - ws] << bs;
                if (bs && q - ws - 1 >= 0) v |= B[q - ws - 1] >> (64 - bs);
                C[q] ^= v;
            }
            if (2 * L <= i) {
                L = i + 1 - L;
                B = T;
                m = 1;
            } else {
                ++m;
            }
        } else {
            ++m;
        }
    }
    // connection polynomial C(x) = 1 + c_1 x + .. + c_L x^L  <->  phi(t) = t^L C(1/t): coefficient of t^(L-j) is c_j
    g_phi_taps.clear();
    if (L != MT_DEG) return;  // leaves the tap list empty: callers report the failure
    for (int j = 1; j <= L; ++j)
        if ((C[j >> 6] >> (j & 63)) & 1u) g_phi_taps.push_back(L - j);
}

// p <- p mod phi for a polynomial of degree < 2 * 19937: clear the high bits from the top, folding each onto the taps
void reduce(Poly &p)
{
    for (int e = 2 * MT_DEG - 1; e >= MT_DEG; --e) {
        if (!p.bit(e)) continue;
        p.flip(e);
        const int sh = e - MT_DEG;
        for (int tap : g_phi_taps) p.flip(tap + sh);
    }
}

void square(Poly &p)  // squaring over GF(2) spreads the bits: coefficient i -> 2 i
{
    Poly r;
    r.clear();
    for (int q = 0; q <= (MT_DEG - 1) >> 6; ++q) {
        uint64_t v = p.w[q];
        while (v) {
            const int b = __builtin_ctzll(v);
            v &= v - 1;
            r.flip(2 * (q * 64 + b));
        }
    }
    p = r;
    reduce(p);
}

std::mutex g_poly_mutex;
std::map<int64_t, std::vector<uint32_t>> g_poly_cache;

}  // namespace

// g(t) = t^J mod phi(t) as 624 little-endian 32-bit words (bit i of the polynomial = bit (i & 31) of word i >> 5).
// Returns nullptr if the characteristic polynomial could not be recovered (never observed; callers then refuse).
const uint32_t *mt_jump_poly(int64_t J)
{
    std::call_once(g_phi_once, compute_phi);
    if (g_phi_taps.empty() || J < 0) return nullptr;
    std::lock_guard<std::mutex> lock(g_poly_mutex);
    auto it = g_poly_cache.find(J);
    if (it != g_poly_cache.end()) return it->second.data();
    Poly p;
    p.clear();
    if (J > 0 && (J & 1) == 0) {  // t^J = (t^(J/2))^2: the lane polynomials t^(2^r blk) come out as one squaring each
        auto half = g_poly_cache.find(J / 2);
        if (half != g_poly_cache.end()) {
            for (int i = 0; i < MT_DEG; ++i)
                if ((half->second[(size_t)(i >> 5)] >> (i & 31)) & 1u) p.flip(i);
            square(p);
            std::vector<uint32_t> out(624, 0u);
            for (int i = 0; i < MT_DEG; ++i)
                if (p.bit(i)) out[(size_t)(i >> 5)] |= 1u << (i & 31);
            return g_poly_cache.emplace(J, std::move(out)).first->second.data();
        }
    }
    p.flip(0);  // 1
    int top = 63;
    while (top > 0 && !((J >> top) & 1)) --top;
    for (int b = top; b >= 0; --b) {
        square(p);
        if ((J >> b) & 1) {  // times t
            for (int q = PW - 1; q > 0; --q) p.w[q] = (p.w[q] << 1) | (p.w[q - 1] >> 63);
            p.w[0] <<= 1;
            reduce(p);
        }
    }
    std::vector<uint32_t> out(624, 0u);
    for (int i = 0; i < MT_DEG; ++i)
        if (p.bit(i)) out[(size_t)(i >> 5)] |= 1u << (i & 31);
    return g_poly_cache.emplace(J, std::move(out)).first->second.data();
}

// window[0..624) <- the 624 stream words that start J words after window[0] (host evaluation of the same XOR
// combination the device kernel k_mt_jump computes).
int mt_jump_window_host(uint32_t *window, int64_t J)
{
    if (J == 0) return ACAV_OK;
    const uint32_t *g = mt_jump_poly(J);
    ACAV_REQUIRE(g, ACAV_ESTATE, "could not derive the MT19937 characteristic polynomial");
    std::vector<uint32_t> X((size_t)MT_DEG + 624 + 8);
    memcpy(X.data(), window, 624 * sizeof(uint32_t));
    for (int m = 624; m < MT_DEG + 624; ++m) X[(size_t)m] = mt_next_word(&X[(size_t)m - 624]);
    uint32_t Y[624];
    memset(Y, 0, sizeof(Y));
    for (int i = 0; i < MT_DEG; ++i) {
        if (!((g[i >> 5] >> (i & 31)) & 1u)) continue;
        const uint32_t *src = &X[(size_t)i];
        for (int k = 0; k < 624; ++k) Y[k] ^= src[k];
    }
    // The low 31 bits of a window's word 0 are not part of the generator state (only its top bit feeds the
    // recurrence), so the combination above carries whatever the source window held there -- not the stream word when
    // the source is a freshly seeded array.  Recover them from the step that produced word 623:
    //   X[J+623] = X[J+396] ^ twist((X[J-1] & UPPER) | (X[J] & LOWER))      (twist is invertible)
    {
        const uint32_t v = Y[623] ^ Y[396];
        const uint32_t b0 = v >> 31;  // the matrix constant has its top bit set, y >> 1 has not
        const uint32_t y = ((v ^ (b0 ? 0x9908b0dfu : 0u)) << 1) | b0;
        Y[0] = (Y[0] & 0x80000000u) | (y & 0x7fffffffu);
    }
    memcpy(window, Y, sizeof(Y));
    return ACAV_OK;
}

}  // namespace acav

// Advance the generator by n draws without producing them (what `for _ in range(n): rng.u32()` leaves behind, in
// O(19937 * 624) word operations whatever n is).
ACAV_EXPORT int acav_rng_jump(acav_rng *rng, int64_t n)
{
    ACAV_REQUIRE(rng && n >= 0, ACAV_EINVAL, "bad argument");
    uint32_t mt[624];
    int idx = 0;
    ACAV_TRY(acav_rng_get_state(rng, mt, &idx));
    const int64_t q = (int64_t)idx + n;  // one past the last skipped draw, counted from mt[0]
    if (q <= 624) return acav_rng_set_state(rng, mt, (int)q);
    const int64_t base = 624 * ((q - 1) / 624);  // the block the sequential generator would hold
    ACAV_TRY(acav::mt_jump_window_host(mt, base));
    return acav_rng_set_state(rng, mt, (int)(q - base));
}
