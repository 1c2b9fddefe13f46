// irec_io.cpp -- the .rec wire format's entropy coder: a C++ restatement of the reference's Cython ArithmeticCoder
// (rec/io/entropy_coding.pyx:19-302, 32-bit-precision integer arithmetic coding with "middle" rescaling), behind the
// C ABI of include/irec.h.  Host code: the reference's coder is CPU code too (its only native component), and the
// streams are a few hundred bits per image.  Bit-for-bit the reference's output (tests/test_rec_io.py pins it against
// the real reference coder, built by the test infrastructure, and against committed golden .rec files).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
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
      if (counts[i] > ((int64_t)1 << 40)) return false;            // (the sum below must not overflow)
      C[i] = c; c += counts[i]; D[i] = c;
      if (c > ((int64_t)1 << 40)) return false;
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
// the decoding loop of irec_ac_decode below; `emit(symbol)` receives every decoded symbol, the terminator 0 last
template <class Emit>
irec_status ac_decode_impl(const Cdf &cdf, const uint8_t *bits, int64_t n_bits, int32_t precision, Emit &&emit) {
  const int32_t n_symbols = (int32_t)cdf.C.size();
  const int64_t whole = (int64_t)1 << precision, half = whole >> 1, quarter = whole >> 2;
  // a model whose total exceeds a quarter of the code range can give a symbol an empty interval (the reference's coder then
  // never terminates): an error here
  if (cdf.R > quarter) return io_fail("irec_ac_decode: the model's total count exceeds 2^(precision - 2)");
  int64_t low = 0, high = whole, z = 0, i = 0, n = 0;
  auto bit = [&](int64_t p) { return p < n_bits && bits[p] == '1'; };
  while (i < precision && i < n_bits) { if (bit(i)) z += (int64_t)1 << (precision - i - 1); ++i; }
  const int64_t max_out = n_bits * 64 + ((int64_t)1 << 22); // guards corrupt input only: a near-deterministic model (max_index 1)
                                                            // legitimately packs thousands of symbols into a bit
  for (;;) {
    const int64_t width = high - low, target = z - low;
    // largest j with (width*C[j])//R <= target (C[0] = 0 always qualifies).  floor(width*C/R) <= target  <=>
    // width*C < (target+1)*R  <=>  C <= ((target+1)*R - 1)//width: ONE division, then a search over the integers C[j]
    // ((target+1)*R <= width*R, the magnitude the reference's own width*D[j] reaches: no new overflow)
    int32_t lo = 0, hi = n_symbols - 1, j = 0;
    if (target < 0 || width <= 0) return io_fail("irec_ac_decode: corrupt stream");
    const int64_t v = ((target + 1) * cdf.R - 1) / width;
    while (lo <= hi) {
      const int32_t mid = (lo + hi) / 2;
      if (cdf.C[mid] <= v) { j = mid; lo = mid + 1; } else hi = mid - 1;
    }
    const int64_t low_ = low + (width * cdf.C[j]) / cdf.R, high_ = low + (width * cdf.D[j]) / cdf.R;
    emit((int64_t)j);
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
  return IREC_OK;
}
} // namespace

extern "C" {

const char *irec_io_last_error(void) { return g_io_error.c_str(); }

// ArithmeticCoder(P, precision).encode(message) -- entropy_coding.pyx:51-121.  out_bits receives one ASCII '0'/'1' per
// code bit (the reference returns a list of such characters); *n_bits is the code length even when it exceeds cap.
irec_status irec_ac_encode(const int64_t *counts, int32_t n_symbols, const int64_t *message, int64_t n_message,
                           int32_t precision, uint8_t *out_bits, int64_t cap, int64_t *n_bits) try {
  Cdf cdf;
  if (!cdf.build(counts, n_symbols)) return io_fail("irec_ac_encode: counts must be >= 1");
  if (precision < 8 || precision > 40 || !n_bits || (n_message > 0 && !message) || (cap > 0 && !out_bits))
    return io_fail("irec_ac_encode: bad arguments");
  const int64_t whole = (int64_t)1 << precision, half = whole >> 1, quarter = whole >> 2;
  if (cdf.R > quarter) return io_fail("irec_ac_encode: the model's total count exceeds 2^(precision - 2)");
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
} catch (const std::exception &e) { g_io_error = std::string("irec_ac_encode: ") + e.what(); return IREC_E_INVALID; }

// ArithmeticCoder.decode_fast(code) -- entropy_coding.pyx:213-302: decodes until the terminator symbol 0.
// The reference finds the symbol with an AVL tree over C (data_structures.py:186-213, the tightest lower bound
// (width*C[j])//R <= z - low); here the same j comes from a binary search.  *n_message is the decoded length
// (terminator included) even when it exceeds cap.
irec_status irec_ac_decode(const int64_t *counts, int32_t n_symbols, const uint8_t *bits, int64_t n_bits,
                           int32_t precision, int64_t *out_message, int64_t cap, int64_t *n_message) try {
  Cdf cdf;
  if (!cdf.build(counts, n_symbols)) return io_fail("irec_ac_decode: counts must be >= 1");
  if (precision < 8 || precision > 40 || !n_message || (n_bits > 0 && !bits) || (cap > 0 && !out_message))
    return io_fail("irec_ac_decode: bad arguments");
  int64_t n = 0;
  if (irec_status st = ac_decode_impl(cdf, bits, n_bits, precision, [&](int64_t j) { if (n < cap) out_message[n] = j; ++n; })) return st;
  *n_message = n;
  if (n > cap) return io_fail("irec_ac_decode: output buffer too small");
  return IREC_OK;
} catch (const std::exception &e) { g_io_error = std::string("irec_ac_decode: ") + e.what(); return IREC_E_INVALID; }

// int('1' + code, 2).to_bytes(ceil((len+1)/8), 'big') -- rec/io/utils.py:66-72,100-106: a leading 1 bit, then the code,
// right-aligned in big-endian bytes.  Returns the byte count (or -1 if cap is too small).
int64_t irec_rec_pack_bits(const uint8_t *bits, int64_t n_bits, uint8_t *out_bytes, int64_t cap) {
  const int64_t total = n_bits + 1, nbytes = (total + 7) / 8;
  if (n_bits < 0 || !out_bytes || cap < nbytes || (n_bits > 0 && !bits)) return -1;
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

// ---- whole .rec files with the default symbol models (rec/io/utils.py:7-106 / :109-216) ------------------------------------
namespace {
// one stream: message = values + 1, terminated by 0 (utils.py:58-68); model = count 1 for the terminator, 1 + weight for every
// other symbol (utils.py:31-35: weight 1000 for indices over max_index + 1 symbols; :41-47: weight 100 for partition counts)
bool encode_stream(const int32_t *values, int64_t n, int32_t n_values, int64_t weight, std::vector<uint8_t> &out) {
  std::vector<int64_t> model((size_t)n_values + 1, 1 + weight), msg((size_t)n + 1);
  model[0] = 1;
  for (int64_t i = 0; i < n; ++i) {
    if (values[i] < 0 || values[i] >= n_values) return false;
    msg[(size_t)i] = (int64_t)values[i] + 1;
  }
  msg[(size_t)n] = 0;
  std::vector<uint8_t> bits((size_t)(64 + 40 * (n + 1)));
  int64_t nb = 0;
  irec_status st = irec_ac_encode(model.data(), n_values + 1, msg.data(), n + 1, 32, bits.data(), (int64_t)bits.size(), &nb);
  if (st != IREC_OK && nb > (int64_t)bits.size()) {
    bits.resize((size_t)nb);
    st = irec_ac_encode(model.data(), n_values + 1, msg.data(), n + 1, 32, bits.data(), (int64_t)bits.size(), &nb);
  }
  if (st != IREC_OK) return false;
  out.resize((size_t)((nb + 1 + 7) / 8));
  return irec_rec_pack_bits(bits.data(), nb, out.data(), (int64_t)out.size()) == (int64_t)out.size();
}
bool decode_stream(const uint8_t *bytes, int64_t n_bytes, int32_t n_values, int64_t weight, std::vector<int32_t> &values) {
  std::vector<int64_t> model((size_t)n_values + 1, 1 + weight);
  model[0] = 1;
  std::vector<uint8_t> bits((size_t)(n_bytes * 8 + 8));
  const int64_t nb = irec_rec_unpack_bits(bytes, n_bytes, bits.data(), (int64_t)bits.size());
  if (nb < 0) return false;
  Cdf cdf;
  if (!cdf.build(model.data(), n_values + 1)) return false;
  values.clear();
  values.reserve((size_t)(nb / 2 + 16));
  bool ended = false;
  const irec_status st = ac_decode_impl(cdf, bits.data(), nb, 32, [&](int64_t j) { if (j == 0) ended = true; else values.push_back((int32_t)(j - 1)); });
  return st == IREC_OK && ended;
}
void put_u32(std::vector<uint8_t> &o, uint32_t v) { for (int k = 0; k < 4; ++k) o.push_back((uint8_t)(v >> (8 * k))); }
void put_u16(std::vector<uint8_t> &o, uint16_t v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); }
uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t get_u16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
} // namespace

// write_compressed_code(file_path, seed, image_shape, block_size, block_indices, max_index) with the default models
// (rec/io/utils.py:7-106), the whole container in one call: static header 'IIIIIHHHH' (:81-90), 4 x R uint32 (:91-95), the R
// partition-count streams, the R index streams (:97-106).  blocks_per_res [R]; K = partitions of every coded block, residual
// block after residual block [sum blocks_per_res]; indices = their sample indices back to back [sum K].
// Returns the file's byte count (also when it exceeds cap: call again), or -1 on invalid input (irec_io_last_error()).
int64_t irec_rec_encode_file(uint32_t seed, uint32_t block_size, uint32_t max_index, uint32_t height, uint32_t width,
                             uint32_t channels, int32_t n_res_blocks, const int32_t *blocks_per_res, const int32_t *K,
                             const int32_t *indices, uint8_t *out, int64_t cap) try {
  if (n_res_blocks < 0 || n_res_blocks > 65535 || (n_res_blocks > 0 && (!blocks_per_res || !K)) || height > 65535 ||
      width > 65535 || channels > 65535) { io_fail("irec_rec_encode_file: bad arguments"); return -1; }
  std::vector<std::vector<uint8_t>> cs((size_t)n_res_blocks), xs((size_t)n_res_blocks);
  std::vector<uint32_t> max_part((size_t)n_res_blocks);
  int64_t kb = 0, ib = 0;
  for (int32_t r = 0; r < n_res_blocks; ++r) {
    const int32_t nb = blocks_per_res[r];
    if (nb < 1) { io_fail("irec_rec_encode_file: a residual block without coded blocks"); return -1; }
    int32_t mx = 0; int64_t tot = 0;
    for (int32_t b = 0; b < nb; ++b) { if (K[kb + b] < 0) { io_fail("irec_rec_encode_file: negative K"); return -1; } mx = K[kb + b] > mx ? K[kb + b] : mx; tot += K[kb + b]; }
    if (tot > 0 && !indices) { io_fail("irec_rec_encode_file: null indices"); return -1; }
    max_part[(size_t)r] = (uint32_t)mx;
    if (!encode_stream(K + kb, nb, mx + 1, 100, cs[(size_t)r])) { io_fail("irec_rec_encode_file: count stream"); return -1; }
    if (!encode_stream(indices + ib, tot, (int32_t)max_index, 1000, xs[(size_t)r])) {
      io_fail("irec_rec_encode_file: an index does not fit max_index (or max_index exceeds what the 32-bit coder's range holds, ~1.07 million)");
      return -1; }   // (the reference overruns its table here)
    kb += nb; ib += tot;
  }
  std::vector<uint8_t> f;
  put_u32(f, seed); put_u32(f, block_size); put_u32(f, max_index); put_u32(f, height); put_u32(f, width);
  put_u16(f, (uint16_t)channels); put_u16(f, 0); put_u16(f, 0); put_u16(f, (uint16_t)n_res_blocks);
  for (int32_t r = 0; r < n_res_blocks; ++r) put_u32(f, (uint32_t)blocks_per_res[r]);
  for (int32_t r = 0; r < n_res_blocks; ++r) put_u32(f, (uint32_t)cs[(size_t)r].size());
  for (int32_t r = 0; r < n_res_blocks; ++r) put_u32(f, (uint32_t)xs[(size_t)r].size());
  for (int32_t r = 0; r < n_res_blocks; ++r) put_u32(f, max_part[(size_t)r]);
  for (auto &v : cs) f.insert(f.end(), v.begin(), v.end());
  for (auto &v : xs) f.insert(f.end(), v.begin(), v.end());
  if (out && cap >= (int64_t)f.size()) std::memcpy(out, f.data(), f.size());
  return (int64_t)f.size();
} catch (const std::exception &e) { g_io_error = std::string("irec_rec_encode_file: ") + e.what(); return -1; }

// read_compressed_code (rec/io/utils.py:109-216) for files written with the default models.  header_out[9] = seed, block_size,
// max_index, height, width, channels, uses_count_file, uses_index_file, R.  Two-call protocol: with null / short outputs it
// returns the sizes in sizes_out[3] = {R, total coded blocks, total indices} and IREC_E_WORKSPACE; with room it fills
// blocks_per_res [R], K [blocks], indices [total].
irec_status irec_rec_decode_file(const uint8_t *bytes, int64_t n_bytes, uint32_t *header_out, int64_t *sizes_out,
                                 int32_t *blocks_per_res, int64_t cap_res, int32_t *K, int64_t cap_blocks, int32_t *indices,
                                 int64_t cap_indices) try {
  if (!bytes || n_bytes < 28 || !header_out || !sizes_out) return io_fail("irec_rec_decode_file: bad arguments");
  for (int k = 0; k < 5; ++k) header_out[k] = get_u32(bytes + 4 * k);
  for (int k = 0; k < 4; ++k) header_out[5 + k] = get_u16(bytes + 20 + 2 * k);
  const int64_t R = header_out[8];
  if (header_out[6] || header_out[7]) return io_fail("irec_rec_decode_file: file uses empirical count tables (not the default models)");
  if (n_bytes < 28 + 16 * R) return io_fail("irec_rec_decode_file: truncated header");
  // a damaged header must come back as an error, not as a symbol table of 2^32 entries: max_index is a sample count
  // (n_samples <= 2^24, irec_params), a residual block's largest partition count at most IREC_MAX_PARTITIONS, and a coded
  // block costs at least one bit of its count stream
  if (header_out[2] < 1u || header_out[2] > (1u << 24)) return io_fail("irec_rec_decode_file: max_index out of range (damaged header?)");
  const uint8_t *dyn = bytes + 28;
  int64_t pos = 28 + 16 * R, n_blocks = 0, n_idx = 0;
  std::vector<std::vector<int32_t>> counts((size_t)R), vals((size_t)R);
  int64_t off_x = pos;
  for (int64_t r = 0; r < R; ++r) off_x += get_u32(dyn + 4 * (R + r));
  for (int64_t r = 0; r < R; ++r) {
    const int64_t nc = get_u32(dyn + 4 * (R + r)), nx = get_u32(dyn + 4 * (2 * R + r));
    const int64_t mx64 = get_u32(dyn + 4 * (3 * R + r));
    // (count model [1, 101, 101, ...] over mx + 2 symbols: with mx >= 1 a block's count costs >= log2(203 / 101) > 1 bit; a
    //  residual block whose K are ALL zero -- mx = 0, model [1, 101] -- costs log2(102 / 101) = 0.0142 bit per block, i.e. up
    //  to 71 blocks per bit of its count stream: the reference writes and reads such files)
    const int64_t blocks_cap = (mx64 >= 1 ? 1 : 72) * (8 * nc + 8);
    if (mx64 > IREC_MAX_PARTITIONS || (int64_t)get_u32(dyn + 4 * r) > blocks_cap)
      return io_fail("irec_rec_decode_file: partition / block counts out of range (damaged header?)");
    const int32_t mx = (int32_t)mx64;
    if (pos + nc > n_bytes || off_x + nx > n_bytes) return io_fail("irec_rec_decode_file: truncated streams");
    if (!decode_stream(bytes + pos, nc, mx + 1, 100, counts[(size_t)r])) return io_fail("irec_rec_decode_file: corrupt count stream");
    if (!decode_stream(bytes + off_x, nx, (int32_t)header_out[2], 1000, vals[(size_t)r])) return io_fail("irec_rec_decode_file: corrupt index stream");
    int64_t tot = 0;
    for (int32_t c : counts[(size_t)r]) tot += c;
    if ((int64_t)counts[(size_t)r].size() != (int64_t)get_u32(dyn + 4 * r) || tot != (int64_t)vals[(size_t)r].size())
      return io_fail("irec_rec_decode_file: streams do not match the header");
    pos += nc; off_x += nx; n_blocks += (int64_t)counts[(size_t)r].size(); n_idx += tot;
  }
  sizes_out[0] = R; sizes_out[1] = n_blocks; sizes_out[2] = n_idx;
  if (!blocks_per_res || !K || (n_idx > 0 && !indices) || cap_res < R || cap_blocks < n_blocks || cap_indices < n_idx) {
    g_io_error = "irec_rec_decode_file: output buffers too small (sizes returned)";
    return IREC_E_WORKSPACE;
  }
  int64_t kb = 0, ib = 0;
  for (int64_t r = 0; r < R; ++r) {
    blocks_per_res[r] = (int32_t)counts[(size_t)r].size();
    for (int32_t c : counts[(size_t)r]) K[kb++] = c;
    for (int32_t v : vals[(size_t)r]) indices[ib++] = v;
  }
  return IREC_OK;
} catch (const std::exception &e) { g_io_error = std::string("irec_rec_decode_file: ") + e.what(); return IREC_E_INVALID; }

// ---- many containers at once (round 3) --------------------------------------------------------------------------------------
// The per-image loop of the reference's evaluation script writes one .rec per image (compression_performance.py:350-365) and
// reads it back (:369-375).  A batched model pass hands over ONE packed read-back for all images:
//     K   [n_images][R][bpt]            partitions of block j of residual block r of image i
//     idx [n_images][R][bpt][max_K]     its index row (the first K entries count)
// irec_rec_encode_files builds the n_images containers from it on `n_threads` host threads (0 = one per core, at most 32):
// out = the files back to back, offsets[n_images + 1].  Each file is byte for byte what irec_rec_encode_file / the
// reference's write_compressed_code produce for that image.  Returns the total byte count (also when it exceeds cap: call
// again with room), -1 on invalid input.
} // extern "C"
namespace {
template <class F>
void for_each_image(int32_t n_images, int32_t n_threads, F &&body) {
  int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  T = T < 1 ? 1 : (T > 32 ? 32 : T);
  if (T > n_images) T = n_images;
  if (T <= 1) { for (int32_t i = 0; i < n_images; ++i) body(i); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t)   // (body() reports through its own error slot and must not throw: an exception leaving a thread ends the process)
    pool.emplace_back([&, t]() { for (int32_t i = t; i < n_images; i += T) body(i); });
  for (auto &th : pool) th.join();
}
} // namespace
extern "C" {

int64_t irec_rec_encode_files(uint32_t seed, uint32_t block_size, uint32_t max_index, uint32_t height, uint32_t width,
                              uint32_t channels, int32_t n_images, int32_t n_res_blocks, int32_t blocks_per_res, int32_t max_K,
                              const int32_t *K, const int32_t *idx, uint8_t *out, int64_t cap, int64_t *offsets, int32_t n_threads) try {
  if (n_images < 0 || n_res_blocks < 1 || blocks_per_res < 1 || max_K < 0 || !K || (max_K > 0 && !idx) || !offsets) {
    io_fail("irec_rec_encode_files: bad arguments"); return -1; }
  std::vector<std::vector<uint8_t>> files((size_t)n_images);
  std::vector<std::string> errs((size_t)n_images);
  const int64_t per_img = (int64_t)n_res_blocks * blocks_per_res;
  const std::vector<int32_t> bpr((size_t)n_res_blocks, blocks_per_res);
  for_each_image(n_images, n_threads, [&](int32_t i) { try {
    const int32_t *Ki = K + (int64_t)i * per_img;
    std::vector<int32_t> flat;
    flat.reserve((size_t)per_img * 8);
    for (int64_t b = 0; b < per_img; ++b) {
      const int32_t k = Ki[b];
      if (k < 0 || k > max_K) { errs[(size_t)i] = "irec_rec_encode_files: K out of range"; return; }
      const int32_t *row = idx + ((int64_t)i * per_img + b) * max_K;
      flat.insert(flat.end(), row, row + k);
    }
    std::vector<uint8_t> &f = files[(size_t)i];
    f.resize((size_t)(64 + 16 * n_res_blocks + 4 * (per_img + (int64_t)flat.size()) + 64));
    int64_t n = irec_rec_encode_file(seed, block_size, max_index, height, width, channels, n_res_blocks, bpr.data(), Ki,
                                     flat.empty() ? nullptr : flat.data(), f.data(), (int64_t)f.size());
    if (n > (int64_t)f.size()) {
      f.resize((size_t)n);
      n = irec_rec_encode_file(seed, block_size, max_index, height, width, channels, n_res_blocks, bpr.data(), Ki,
                               flat.empty() ? nullptr : flat.data(), f.data(), (int64_t)f.size());
    }
    if (n < 0) { errs[(size_t)i] = g_io_error; return; }
    f.resize((size_t)n);
  } catch (const std::exception &e) { errs[(size_t)i] = std::string("irec_rec_encode_files: ") + e.what(); } });
  int64_t total = 0;
  for (int32_t i = 0; i < n_images; ++i) {
    if (!errs[(size_t)i].empty()) { g_io_error = errs[(size_t)i] + " (image " + std::to_string(i) + ")"; return -1; }
    offsets[i] = total; total += (int64_t)files[(size_t)i].size();
  }
  offsets[n_images] = total;
  if (out && cap >= total)
    for (int32_t i = 0; i < n_images; ++i) std::memcpy(out + offsets[i], files[(size_t)i].data(), files[(size_t)i].size());
  return total;
} catch (const std::exception &e) { g_io_error = std::string("irec_rec_encode_files: ") + e.what(); return -1; }

// The inverse: n_images containers (bytes at offsets[i] .. offsets[i + 1]) decoded on host threads into the packed layout
// above (rows zero-filled past K).  headers [n_images][9] as irec_rec_decode_file's; every file must hold n_res_blocks
// residual blocks of blocks_per_res coded blocks with at most max_K partitions, else the call fails naming the image.
irec_status irec_rec_decode_files(const uint8_t *bytes, const int64_t *offsets, int32_t n_images, int32_t n_res_blocks,
                                  int32_t blocks_per_res, int32_t max_K, uint32_t *headers, int32_t *K, int32_t *idx,
                                  int32_t n_threads) try {
  if (!bytes || !offsets || n_images < 0 || n_res_blocks < 1 || blocks_per_res < 1 || max_K < 0 || !headers || !K || (max_K > 0 && !idx))
    return io_fail("irec_rec_decode_files: bad arguments");
  std::vector<std::string> errs((size_t)n_images);
  const int64_t per_img = (int64_t)n_res_blocks * blocks_per_res;
  for_each_image(n_images, n_threads, [&](int32_t i) { try {
    std::vector<int32_t> bpr((size_t)n_res_blocks), flat;
    int64_t sizes[3] = {0, 0, 0};
    int32_t *Ki = K + (int64_t)i * per_img;
    const uint8_t *p = bytes + offsets[i];
    const int64_t nb = offsets[i + 1] - offsets[i];
    // an index costs at least ~1.4 bits under the default model unless max_index is tiny: size the buffer from the file, retry once
    flat.resize((size_t)(8 * nb + 1024));
    irec_status st = IREC_E_WORKSPACE;
    for (int attempt = 0; attempt < 2 && st == IREC_E_WORKSPACE; ++attempt) {
      std::vector<int32_t> Ktmp((size_t)std::max<int64_t>(per_img, sizes[1]));
      st = irec_rec_decode_file(p, nb, headers + 9 * (int64_t)i, sizes, bpr.data(), n_res_blocks, Ktmp.data(), (int64_t)Ktmp.size(),
                                flat.data(), (int64_t)flat.size());
      if (st == IREC_E_WORKSPACE) {
        if (sizes[0] != n_res_blocks || sizes[1] != per_img) { st = IREC_E_INVALID; g_io_error = "irec_rec_decode_files: block structure differs"; break; }
        flat.resize((size_t)sizes[2] + 1);
      } else if (st == IREC_OK) {
        if (sizes[0] != n_res_blocks || sizes[1] != per_img) { st = IREC_E_INVALID; g_io_error = "irec_rec_decode_files: block structure differs"; break; }
        for (int32_t r = 0; r < n_res_blocks; ++r)
          if (bpr[(size_t)r] != blocks_per_res) { st = IREC_E_INVALID; g_io_error = "irec_rec_decode_files: block structure differs"; }
        if (st != IREC_OK) break;
        int64_t at = 0;
        for (int64_t b = 0; b < per_img; ++b) {
          const int32_t k = Ktmp[(size_t)b];
          if (k > max_K) { st = IREC_E_INVALID; g_io_error = "irec_rec_decode_files: more partitions than max_K"; break; }
          Ki[b] = k;
          int32_t *row = idx + ((int64_t)i * per_img + b) * max_K;
          for (int32_t t = 0; t < max_K; ++t) row[t] = t < k ? flat[(size_t)(at + t)] : 0;
          at += k;
        }
      }
    }
    if (st != IREC_OK) errs[(size_t)i] = g_io_error.empty() ? std::string("irec_rec_decode_files: failed") : g_io_error;
  } catch (const std::exception &e) { errs[(size_t)i] = std::string("irec_rec_decode_files: ") + e.what(); } });
  for (int32_t i = 0; i < n_images; ++i)
    if (!errs[(size_t)i].empty()) { g_io_error = errs[(size_t)i] + " (image " + std::to_string(i) + ")"; return IREC_E_INVALID; }
  return IREC_OK;
} catch (const std::exception &e) { g_io_error = std::string("irec_rec_decode_files: ") + e.what(); return IREC_E_INVALID; }

} // extern "C"
