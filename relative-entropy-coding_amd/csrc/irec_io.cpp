// irec_io.cpp -- the .rec wire format's entropy coder: a C++ restatement of the reference's Cython ArithmeticCoder
// (rec/io/entropy_coding.pyx:19-302, 32-bit-precision integer arithmetic coding with "middle" rescaling), behind the
// C ABI of include/irec.h.  Host code: the reference's coder is CPU code too (its only native component), and the
// streams are a few hundred bits per image.  Bit-for-bit the reference's output (tests/test_rec_io.py pins it against
// the real reference coder, built by the test infrastructure, and against committed golden .rec files).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "irec.h"

namespace {
thread_local std::string g_io_error;
irec_status io_fail(const char *msg) { g_io_error = msg; return IREC_E_INVALID; }

struct Cdf {
  std::vector<int64_t> C, D; // cumulative low / high of every symbol (entropy_coding.pyx:27-46)
  int64_t R = 0;
  bool build(const int64_t *counts, int32_t n) {
    if (!counts || n < 1) return false;
    C.resize(n); D.resize(n);
    int64_t c = 0;
    for (int32_t i = 0; i < n; ++i) {
      if (counts[i] < 1) return false; // strictly increasing C is what the reference's interval tree assumes
      C[i] = c; c += counts[i]; D[i] = c;
    }
    R = c;
    return true;
  }
};

struct BitSink {
  uint8_t *out; int64_t cap, n = 0; bool overflow = false;
  void put(char bit, int64_t follow) { // code.extend(bit + other * follow), entropy_coding.pyx:88-100
    push(bit);
    const char other = bit == '0' ? '1' : '0';
    for (int64_t i = 0; i < follow; ++i) push(other);
  }
  void push(char b) { if (n < cap) out[n] = (uint8_t)b; else overflow = true; ++n; }
};
} // namespace

extern "C" {

const char *irec_io_last_error(void) { return g_io_error.c_str(); }

// ArithmeticCoder(P, precision).encode(message) -- entropy_coding.pyx:51-121.  out_bits receives one ASCII '0'/'1' per
// code bit (the reference returns a list of such characters); *n_bits is the code length even when it exceeds cap.
irec_status irec_ac_encode(const int64_t *counts, int32_t n_symbols, const int64_t *message, int64_t n_message,
                           int32_t precision, uint8_t *out_bits, int64_t cap, int64_t *n_bits) {
  Cdf cdf;
  if (!cdf.build(counts, n_symbols)) return io_fail("irec_ac_encode: counts must be >= 1");
  if (precision < 8 || precision > 40 || !n_bits || (n_message > 0 && !message) || (cap > 0 && !out_bits))
    return io_fail("irec_ac_encode: bad arguments");
  const int64_t whole = (int64_t)1 << precision, half = whole >> 1, quarter = whole >> 2;
  int64_t low = 0, high = whole, s = 0;
  BitSink sink{out_bits, cap};
  for (int64_t k = 0; k < n_message; ++k) {
    const int64_t m = message[k];
    if (m < 0 || m >= n_symbols) return io_fail("irec_ac_encode: symbol out of range");
    const int64_t width = high - low;
    high = low + (width * cdf.D[m]) / cdf.R;
    low = low + (width * cdf.C[m]) / cdf.R;
    while (high < half || low > half) {           // interval subdivision
      if (high < half) { sink.put('0', s); s = 0; low *= 2; high *= 2; }
      else { sink.put('1', s); s = 0; low = (low - half) * 2; high = (high - half) * 2; }
    }
    while (low > quarter && high < 3 * quarter) { // middle rescaling
      s += 1; low = (low - quarter) * 2; high = (high - quarter) * 2;
    }
  }
  s += 1;                                         // final emission
  if (low <= quarter) sink.put('0', s); else sink.put('1', s);
  *n_bits = sink.n;
  if (sink.overflow) return io_fail("irec_ac_encode: output buffer too small");
  return IREC_OK;
}

// ArithmeticCoder.decode_fast(code) -- entropy_coding.pyx:213-302: decodes until the terminator symbol 0.
// The reference finds the symbol with an AVL tree over C (data_structures.py:186-213, the tightest lower bound
// (width*C[j])//R <= z - low); here the same j comes from a binary search.  *n_message is the decoded length
// (terminator included) even when it exceeds cap.
irec_status irec_ac_decode(const int64_t *counts, int32_t n_symbols, const uint8_t *bits, int64_t n_bits,
                           int32_t precision, int64_t *out_message, int64_t cap, int64_t *n_message) {
  Cdf cdf;
  if (!cdf.build(counts, n_symbols)) return io_fail("irec_ac_decode: counts must be >= 1");
  if (precision < 8 || precision > 40 || !n_message || (n_bits > 0 && !bits) || (cap > 0 && !out_message))
    return io_fail("irec_ac_decode: bad arguments");
  const int64_t whole = (int64_t)1 << precision, half = whole >> 1, quarter = whole >> 2;
  int64_t low = 0, high = whole, z = 0, i = 0, n = 0;
  auto bit = [&](int64_t p) { return p < n_bits && bits[p] == '1'; };
  while (i < precision && i < n_bits) { if (bit(i)) z += (int64_t)1 << (precision - i - 1); ++i; }
  const int64_t max_out = n_bits * 64 + 64; // a valid stream ends long before; guards corrupt input
  for (;;) {
    const int64_t width = high - low, target = z - low;
    int32_t lo = 0, hi = n_symbols - 1, j = 0; // largest j with (width*C[j])//R <= target; C[0] = 0 always qualifies
    if (target < 0) return io_fail("irec_ac_decode: corrupt stream");
    while (lo <= hi) {
      const int32_t mid = (lo + hi) / 2;
      if ((width * cdf.C[mid]) / cdf.R <= target) { j = mid; lo = mid + 1; } else hi = mid - 1;
    }
    const int64_t low_ = low + (width * cdf.C[j]) / cdf.R, high_ = low + (width * cdf.D[j]) / cdf.R;
    if (n < cap) out_message[n] = j;
    ++n;
    high = high_; low = low_;
    if (j == 0) break;
    if (n > max_out) return io_fail("irec_ac_decode: no terminator found");
    while (high < half || low > half) {
      if (high < half) { low *= 2; high *= 2; z *= 2; }
      else { low = (low - half) * 2; high = (high - half) * 2; z = (z - half) * 2; }
      if (bit(i)) z += 1;
      ++i;
    }
    while (low > quarter && high < 3 * quarter) {
      low = (low - quarter) * 2; high = (high - quarter) * 2; z = (z - quarter) * 2;
      if (bit(i)) z += 1;
      ++i;
    }
  }
  *n_message = n;
  if (n > cap) return io_fail("irec_ac_decode: output buffer too small");
  return IREC_OK;
}

// int('1' + code, 2).to_bytes(ceil((len+1)/8), 'big') -- rec/io/utils.py:66-72,100-106: a leading 1 bit, then the code,
// right-aligned in big-endian bytes.  Returns the byte count (or -1 if cap is too small).
int64_t irec_rec_pack_bits(const uint8_t *bits, int64_t n_bits, uint8_t *out_bytes, int64_t cap) {
  const int64_t total = n_bits + 1, nbytes = (total + 7) / 8;
  if (!out_bytes || cap < nbytes || (n_bits > 0 && !bits)) return -1;
  std::memset(out_bytes, 0, (size_t)nbytes);
  const int64_t pad = nbytes * 8 - total; // leading zero bits
  for (int64_t p = 0; p < total; ++p) {
    const bool one = p == 0 ? true : bits[p - 1] == '1';
    if (one) { const int64_t q = pad + p; out_bytes[q >> 3] |= (uint8_t)(0x80u >> (q & 7)); }
  }
  return nbytes;
}

// bin(int.from_bytes(b, 'big'))[3:] -- rec/io/utils.py:158-170: strips leading zeros and the marker 1 bit.
// Returns the number of code bits written as ASCII '0'/'1' (or -1).
int64_t irec_rec_unpack_bits(const uint8_t *bytes, int64_t n_bytes, uint8_t *out_bits, int64_t cap) {
  if (n_bytes > 0 && !bytes) return -1;
  int64_t first = -1;
  for (int64_t q = 0; q < n_bytes * 8 && first < 0; ++q)
    if (bytes[q >> 3] & (0x80u >> (q & 7))) first = q;
  if (first < 0) return -1; // no marker bit
  const int64_t n = n_bytes * 8 - first - 1;
  if (n > cap || (n > 0 && !out_bits)) return -1;
  for (int64_t p = 0; p < n; ++p) {
    const int64_t q = first + 1 + p;
    out_bits[p] = (bytes[q >> 3] & (0x80u >> (q & 7))) ? '1' : '0';
  }
  return n;
}

} // extern "C"
