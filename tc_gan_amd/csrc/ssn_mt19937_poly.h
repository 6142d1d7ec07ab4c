// Jump-ahead polynomials of MT19937 (host arithmetic in GF(2)[t]; no HIP in this header).
//
// The reference draws its noise from numpy's RandomState = MT19937 (tc_gan/networks/ssn.py:434-439); the device generator of
// ssn_mt19937.hip starts its segments from states far ahead of the caller's, and gets them by the polynomial method
// (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer 2008): with x[0..624) the state words and x[k+624] = x[k+397] ^
// twist(x[k], x[k+1]) the untempered word sequence, every bit column of x obeys the linear recurrence whose characteristic
// polynomial phi (degree 19937) is computed here by Berlekamp-Massey from the generator's own output, and for
// g = t^J mod phi the state J words ahead is  word j = XOR over the set bits i of g of x[i + j]  (bit 31 of word 0 and
// words 1..623: the 19937 state bits).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace ssn { namespace mt {

constexpr int kN = 624, kM = 397, kDeg = 19937;
constexpr int kPW = 313;           // 64-bit words of a polynomial of degree <= 19937 (with room up to bit 20031)
constexpr int kPW2 = 2 * kPW;      // a product before reduction

struct Poly { uint64_t w[kPW]; };

inline uint32_t twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
inline bool get_bit(const uint64_t* w, int i) { return (w[i >> 6] >> (i & 63)) & 1u; }
inline void flip_bit(uint64_t* w, int i) { w[i >> 6] ^= (uint64_t)1 << (i & 63); }

// dst[0..n+1) ^= src[0..n) << sh  (0 <= sh < 64), dst offset by whole words by the caller
inline void xor_shifted(uint64_t* dst, const uint64_t* src, int n, int sh) {
    if (sh == 0) { for (int k = 0; k < n; ++k) dst[k] ^= src[k]; return; }
    uint64_t carry = 0;
    for (int k = 0; k < n; ++k) {
        dst[k] ^= (src[k] << sh) | carry;
        carry = src[k] >> (64 - sh);
    }
    dst[n] ^= carry;
}

class Field {
public:
    // phi by Berlekamp-Massey over one bit column of the generator's word sequence; false if the result is not of degree 19937
    bool init() {
        // bit sequence: bit 0 of x[1], x[2], ... from the state of init_genrand(5489) (any state that is not all zero will do)
        const int len = 2 * kDeg + 64;
        std::vector<uint32_t> x(len + kN + 1);
        uint32_t s = 5489u;
        for (int i = 0; i < kN; ++i) { x[i] = s; s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u; }
        for (int k = 0; k + kN < (int)x.size(); ++k) x[k + kN] = x[k + kM] ^ twist(x[k], x[k + 1]);
        std::vector<uint64_t> C(kPW + 1, 0), B(kPW + 1, 0), T(kPW + 1, 0), R(kPW + 1, 0);
        C[0] = B[0] = 1;
        int L = 0, m = 1;
        for (int n = 0; n < len; ++n) {
            // R bit i = s[n - i]
            uint64_t carry = x[n + 1] & 1u;
            for (int k = 0; k < kPW; ++k) { const uint64_t nc = R[k] >> 63; R[k] = (R[k] << 1) | carry; carry = nc; }
            uint64_t acc = 0;
            for (int k = 0; k < kPW; ++k) acc ^= C[k] & R[k];
            if (!(__builtin_popcountll(acc) & 1)) { ++m; continue; }
            if (2 * L <= n) {
                T = C;
                if (m < 64 * kPW) xor_shifted(C.data() + (m >> 6), B.data(), kPW - (m >> 6), m & 63);
                L = n + 1 - L; B = T; m = 1;
            } else {
                if (m < 64 * kPW) xor_shifted(C.data() + (m >> 6), B.data(), kPW - (m >> 6), m & 63);
                ++m;
            }
            if (L > kDeg) return false;
        }
        if (L != kDeg) return false;
        // connection polynomial C (s[n] = sum C_i s[n-i]) -> characteristic polynomial phi_i = C_{L-i}
        std::memset(phi_.w, 0, sizeof phi_.w);
        for (int i = 0; i <= kDeg; ++i) if (get_bit(C.data(), kDeg - i)) flip_bit(phi_.w, i);
        if (!get_bit(phi_.w, kDeg) || !get_bit(phi_.w, 0)) return false;
        for (int r = 0; r < 64; ++r) {
            std::memset(phis_[r], 0, sizeof phis_[r]);
            xor_shifted(phis_[r], phi_.w, kPW, r);
        }
        ok_ = true;
        return true;
    }
    bool ok() const { return ok_; }
    const Poly& phi() const { return phi_; }

    // c (kPW2 words, degree < 2 * 19937) -> c mod phi, in place; the low kPW words hold the result
    void reduce(uint64_t* c) const {
        for (int k = 2 * kDeg; k >= kDeg; --k) {
            if (!get_bit(c, k)) continue;
            const int sh = k - kDeg;
            uint64_t* d = c + (sh >> 6);
            const uint64_t* p = phis_[sh & 63];
            for (int q = 0; q < kPW + 1; ++q) d[q] ^= p[q];
        }
    }
    Poly mul(const Poly& a, const Poly& b) const {
        std::vector<uint64_t> c(kPW2 + 2, 0);
        static thread_local std::vector<uint64_t> bs;          // b << r, r = 0..63
        bs.assign((size_t)64 * (kPW + 1), 0);
        for (int r = 0; r < 64; ++r) xor_shifted(bs.data() + (size_t)r * (kPW + 1), b.w, kPW, r);
        for (int wi = 0; wi < kPW; ++wi) {
            uint64_t bits = a.w[wi];
            while (bits) {
                const int r = __builtin_ctzll(bits);
                bits &= bits - 1;
                const uint64_t* p = bs.data() + (size_t)r * (kPW + 1);
                uint64_t* d = c.data() + wi;
                for (int q = 0; q < kPW + 1; ++q) d[q] ^= p[q];
            }
        }
        reduce(c.data());
        Poly out;
        std::memcpy(out.w, c.data(), sizeof out.w);
        return out;
    }
    Poly sqr(const Poly& a) const {
        std::vector<uint64_t> c(kPW2 + 2, 0);
        for (int wi = 0; wi < kPW; ++wi) {
            c[2 * wi] = spread((uint32_t)a.w[wi]);
            c[2 * wi + 1] = spread((uint32_t)(a.w[wi] >> 32));
        }
        reduce(c.data());
        Poly out;
        std::memcpy(out.w, c.data(), sizeof out.w);
        return out;
    }
    static Poly monomial(int e) {          // t^e, e <= 19936
        Poly p;
        std::memset(p.w, 0, sizeof p.w);
        flip_bit(p.w, e);
        return p;
    }
    // t^(624 * nblocks) mod phi
    Poly block_jump(unsigned long long nblocks) const {
        Poly result = monomial(0), base = monomial(kN);
        while (nblocks) {
            if (nblocks & 1) result = mul(result, base);
            nblocks >>= 1;
            if (nblocks) base = sqr(base);
        }
        return result;
    }

private:
    static uint64_t spread(uint32_t v) {        // bit i -> bit 2 i
        uint64_t x = v;
        x = (x | (x << 16)) & 0x0000ffff0000ffffull;
        x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
        x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
        x = (x | (x << 2)) & 0x3333333333333333ull;
        x = (x | (x << 1)) & 0x5555555555555555ull;
        return x;
    }
    Poly phi_;
    uint64_t phis_[64][kPW + 2];
    bool ok_ = false;
};

}}  // namespace ssn::mt
