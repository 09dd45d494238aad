// The Fiat-Shamir transcript of the prover (halo2_proofs/src/transcript.rs:150-300) as a host-side object of the library:
// Blake2bWrite<_, C, Challenge255<C>> -- BLAKE2b-512 with the "Halo2-Transcript" personalisation; points absorbed as
// (x, y) canonical little-endian and written compressed, scalars canonical little-endian; a challenge is the digest of a COPY
// of the state, reduced from 512 bits into the scalar field (Challenge255::new, :286-296).
//
// It lives here so that the round loop of the opening argument (h2_ipa_rounds_device) can talk to a transcript without leaving
// native code: h2_transcript_cb_write_point / _cb_squeeze have the loop's callback signatures.  Pure host arithmetic (a few
// hundred bytes hashed per proof): no device is needed and none is touched.
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"
#include "host_field.h"

namespace h2 {

// RFC 7693, unkeyed, 64-byte digest, personalisation in parameter-block bytes 48..63
struct Blake2b {
    u64 h[8], t = 0;
    uint8_t buf[128];
    size_t fill = 0;
    explicit Blake2b(const char personal[16]) {
        static const u64 IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                  0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        u64 pw[2];
        memcpy(pw, personal, 16);
        for (int i = 0; i < 8; i++) h[i] = IV[i];
        h[0] ^= 0x01010000ULL ^ 64;
        h[6] ^= pw[0];
        h[7] ^= pw[1];
    }
    void update(const uint8_t *in, size_t len) {
        while (len) {
            if (fill == 128) {            // compress only once more input is known to follow: the last block is special
                t += 128;
                compress(false);
                fill = 0;
            }
            const size_t take = len < 128 - fill ? len : 128 - fill;
            memcpy(buf + fill, in, take);
            fill += take;
            in += take;
            len -= take;
        }
    }
    void digest(uint8_t out[64]) const {  // of a copy: the transcript keeps absorbing (transcript.rs:202)
        Blake2b c = *this;
        c.t += c.fill;
        memset(c.buf + c.fill, 0, 128 - c.fill);
        c.compress(true);
        memcpy(out, c.h, 64);
    }

  private:
    static u64 rotr(u64 x, int r) { return (x >> r) | (x << (64 - r)); }
    void compress(bool last) {
        static const uint8_t S[12][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
                                          {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
                                          {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
                                          {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
                                          {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
                                          {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        static const u64 IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                  0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        u64 m[16], v[16];
        memcpy(m, buf, 128);
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[8 + i] = IV[i];
        }
        v[12] ^= t;
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, u64 x, u64 y) {
            v[a] = v[a] + v[b] + x;
            v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];
            v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y;
            v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];
            v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t *s = S[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);
            G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);
            G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);
            G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]);
            G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
    }
};

struct Transcript {
    std::mutex mu;
    int curve, bf, sf;
    Blake2b state;
    std::vector<uint8_t> written;
    explicit Transcript(int c) : curve(c), bf(c == H2_PALLAS ? H2_FP : H2_FQ), sf(c == H2_PALLAS ? H2_FQ : H2_FP), state("Halo2-Transcript") {}
};
static std::mutex g_tr_mu;
static std::map<h2_transcript_t, std::shared_ptr<Transcript>> g_tr;
static h2_transcript_t g_tr_next = 1;

static std::shared_ptr<Transcript> find_transcript(h2_transcript_t t) {
    std::lock_guard<std::mutex> lk(g_tr_mu);
    auto it = g_tr.find(t);
    return it == g_tr.end() ? nullptr : it->second;
}

// affine (jacobian == 0: 8 limbs) or Jacobian (12 limbs: the prover's .to_affine() before write_point, prover.rs:116-117),
// Montgomery -> canonical x, y; false for the identity, which a transcript refuses (transcript.rs:209-214)
static bool canonical_xy(int bf, const u64 *p, int jacobian, u64 x[4], u64 y[4]) {
    if (!jacobian) {
        if (host_is_zero(p) && host_is_zero(p + 4)) return false;
        host_from_mont(bf, x, p);
        host_from_mont(bf, y, p + 4);
        return true;
    }
    if (host_is_zero(p + 8)) return false;
    u64 zi[4], zi2[4], zi3[4], ax[4], ay[4];
    host_inv(bf, zi, p + 8);
    host_mul(bf, zi2, zi, zi);
    host_mul(bf, zi3, zi2, zi);
    host_mul(bf, ax, p, zi2);
    host_mul(bf, ay, p + 4, zi3);
    host_from_mont(bf, x, ax);
    host_from_mont(bf, y, ay);
    return true;
}

static int absorb_point(Transcript &T, const u64 *p, int jacobian, bool write) {
    u64 x[4], y[4];
    if (!canonical_xy(T.bf, p, jacobian, x, y)) {
        set_last_error_msg("cannot write points at infinity to the transcript");
        return H2_ERR_ARGS;
    }
    const uint8_t prefix = 1;                                                   // transcript.rs:14-20
    T.state.update(&prefix, 1);
    T.state.update(reinterpret_cast<const uint8_t *>(x), 32);
    T.state.update(reinterpret_cast<const uint8_t *>(y), 32);
    if (write) {
        uint8_t enc[32];
        memcpy(enc, x, 32);
        enc[31] |= (uint8_t)((y[0] & 1) << 7);                                  // pasta_curves to_bytes: the sign of y in the top bit
        T.written.insert(T.written.end(), enc, enc + 32);
    }
    return H2_OK;
}

static int absorb_scalar(Transcript &T, const u64 *s, bool write) {
    u64 c[4];
    host_from_mont(T.sf, c, s);
    const uint8_t prefix = 2;
    T.state.update(&prefix, 1);
    T.state.update(reinterpret_cast<const uint8_t *>(c), 32);
    if (write) T.written.insert(T.written.end(), reinterpret_cast<const uint8_t *>(c), reinterpret_cast<const uint8_t *>(c) + 32);
    return H2_OK;
}

static void squeeze(Transcript &T, u64 out[4]) {
    const uint8_t prefix = 0;
    T.state.update(&prefix, 1);
    uint8_t d[64];
    T.state.digest(d);
    u64 lo[4], hi[4], lm[4], hm[4];
    memcpy(lo, d, 32);
    memcpy(hi, d + 32, 32);
    // from_bytes_wide: (lo + hi 2^256) mod q.  r2 read as a Montgomery element IS 2^256; any raw value below 2^256 may enter host_mul
    host_mul(T.sf, lm, lo, kHostField[T.sf].r2);
    host_mul(T.sf, hm, hi, kHostField[T.sf].r2);
    host_mul(T.sf, hm, hm, kHostField[T.sf].r2);
    host_add(T.sf, out, lm, hm);
}

}  // namespace h2

using namespace h2;

extern "C" int h2_transcript_new(int curve, h2_transcript_t *t) {
    if ((curve != H2_PALLAS && curve != H2_VESTA) || !t) return H2_ERR_ARGS;
    auto tr = std::make_shared<Transcript>(curve);
    std::lock_guard<std::mutex> lk(g_tr_mu);
    *t = g_tr_next++;
    g_tr[*t] = tr;
    return H2_OK;
}

extern "C" int h2_transcript_free(h2_transcript_t t) {
    std::lock_guard<std::mutex> lk(g_tr_mu);
    return g_tr.erase(t) ? H2_OK : H2_ERR_HANDLE;
}

extern "C" int h2_transcript_common_point(h2_transcript_t t, const uint64_t *point, int jacobian) {
    auto T = find_transcript(t);
    if (!T) return H2_ERR_HANDLE;
    if (!point) return H2_ERR_ARGS;
    std::lock_guard<std::mutex> lk(T->mu);
    return absorb_point(*T, point, jacobian, false);
}

extern "C" int h2_transcript_write_point(h2_transcript_t t, const uint64_t *point, int jacobian) {
    auto T = find_transcript(t);
    if (!T) return H2_ERR_HANDLE;
    if (!point) return H2_ERR_ARGS;
    std::lock_guard<std::mutex> lk(T->mu);
    return absorb_point(*T, point, jacobian, true);
}

extern "C" int h2_transcript_common_scalar(h2_transcript_t t, const uint64_t *scalar) {
    auto T = find_transcript(t);
    if (!T) return H2_ERR_HANDLE;
    if (!scalar) return H2_ERR_ARGS;
    std::lock_guard<std::mutex> lk(T->mu);
    return absorb_scalar(*T, scalar, false);
}

extern "C" int h2_transcript_write_scalar(h2_transcript_t t, const uint64_t *scalar) {
    auto T = find_transcript(t);
    if (!T) return H2_ERR_HANDLE;
    if (!scalar) return H2_ERR_ARGS;
    std::lock_guard<std::mutex> lk(T->mu);
    return absorb_scalar(*T, scalar, true);
}

extern "C" int h2_transcript_squeeze_challenge(h2_transcript_t t, uint64_t *challenge) {
    auto T = find_transcript(t);
    if (!T) return H2_ERR_HANDLE;
    if (!challenge) return H2_ERR_ARGS;
    std::lock_guard<std::mutex> lk(T->mu);
    squeeze(*T, challenge);
    return H2_OK;
}

extern "C" int h2_transcript_bytes(h2_transcript_t t, uint8_t *out, size_t cap, size_t *len) {
    auto T = find_transcript(t);
    if (!T) return H2_ERR_HANDLE;
    if (!len || (cap && !out)) return H2_ERR_ARGS;
    std::lock_guard<std::mutex> lk(T->mu);
    *len = T->written.size();
    if (cap == 0) return H2_OK;                                      // size query
    if (cap < T->written.size()) return H2_ERR_ARGS;                 // too small a buffer: *len says how much is needed, nothing is copied
    if (!T->written.empty()) memcpy(out, T->written.data(), T->written.size());
    return H2_OK;
}

// the two callbacks of h2_ipa_rounds_device over a transcript of this library: user = (void *)(uintptr_t)handle
extern "C" int h2_transcript_cb_write_point(void *user, const uint64_t *xy) {
    return h2_transcript_write_point((h2_transcript_t)(uintptr_t)user, xy, 0);
}

extern "C" int h2_transcript_cb_squeeze(void *user, uint64_t *challenge) {
    return h2_transcript_squeeze_challenge((h2_transcript_t)(uintptr_t)user, challenge);
}
