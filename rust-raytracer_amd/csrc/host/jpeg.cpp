// jpeg.cpp — Huffman JPEG -> RGB8: baseline / extended sequential (SOF0, SOF1), interleaved or one component per
// scan, and progressive (SOF2: spectral selection + successive approximation), 8 bits, 1 or 3 components.
// Stands in for the `jpeg-decoder` crate the reference calls while deserializing textures
// (reference materials.rs:213-219 load_texture_image, config.rs:36-47).  Texel values of a
// lossy decode are decoder-specific (IDCT + chroma upsampling), and no reference test pins
// them; what matters for parity is that oracle and GPU consume the SAME decoded buffer,
// which they do (both receive RtTexture.rgb8 from this file).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <new>
#include <vector>

#include "../../../include/rt_abi.h"

namespace {

struct Huff {
  // canonical decode tables
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  int32_t mincode[17], maxcode[18], valptr[17];
  uint8_t look_nbits[512];  // 9-bit fast lookup
  uint8_t look_sym[512];
  bool present = false;
  // false: the code lengths over-subscribe the code space (more than 2^l codes of length <= l) — such a
  // table would index past the 9-bit lookup below; jpeg-decoder rejects it too
  bool build() {
    int code = 0, k = 0;
    std::memset(look_nbits, 0, sizeof look_nbits);
    present = false;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      if (code + int(bits[l]) > (1 << l)) return false;
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        if (l <= 9) {
          int base = code << (9 - l);
          for (int f = 0; f < (1 << (9 - l)); ++f) { look_nbits[base + f] = uint8_t(l); look_sym[base + f] = vals[k]; }
        }
      }
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    present = true;
    return true;
  }
};

struct Comp {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int bw = 0, bh = 0;    // blocks per line / column (padded to MCU)
  int nbx = 0, nby = 0;  // blocks that cover the component itself (what a scan of this component alone walks)
  int pred = 0;
  std::vector<int16_t> coef;   // bw*bh blocks x 64 coefficients, natural order: every scan adds to them
  std::vector<uint8_t> plane;  // bw*8 x bh*8, after the last scan
};

struct BitReader {
  const uint8_t* p; const uint8_t* end;
  uint32_t acc = 0; int cnt = 0; bool hit_marker = false;
  void fill() {
    while (cnt <= 24) {
      uint32_t byte = 0;
      if (!hit_marker && p < end) {
        byte = *p;
        if (byte == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) { p += 2; }
          else { hit_marker = true; byte = 0; }  // leave the marker in place, feed zeros
        } else ++p;
      }
      acc |= byte << (24 - cnt);
      cnt += 8;
    }
  }
  inline uint32_t peek(int n) { if (cnt < n) fill(); return acc >> (32 - n); }
  inline void skip(int n) { acc <<= n; cnt -= n; }
  inline int get(int n) { if (!n) return 0; uint32_t v = peek(n); skip(n); return int(v); }
  void reset() { acc = 0; cnt = 0; hit_marker = false; }
};

inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

int decode_sym(BitReader& br, const Huff& h) {
  uint32_t look = br.peek(9);
  int nb = h.look_nbits[look];
  if (nb) { br.skip(nb); return h.look_sym[look]; }
  uint32_t bits16 = br.peek(16);
  for (int l = 10; l <= 16; ++l) {
    int code = int(bits16 >> (16 - l));
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) {
      br.skip(l);
      return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
  }
  return -1;
}

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// separable 8x8 inverse DCT in double precision (reference-quality, not speed-tuned)
struct IdctTable {
  double c[8][8];  // c[x][u] = 0.5 * C(u) * cos((2x+1) u pi / 16)
  IdctTable() {
    for (int x = 0; x < 8; ++x)
      for (int u = 0; u < 8; ++u)
        c[x][u] = 0.5 * (u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
  }
};
const IdctTable kIdct;

void idct_block(const int* coef, uint8_t* out, int stride) {
  double tmp[64];
  for (int v = 0; v < 8; ++v) {  // rows: over u
    const int* row = coef + v * 8;
    bool ac = false;
    for (int u = 1; u < 8; ++u) ac |= row[u] != 0;
    for (int x = 0; x < 8; ++x) {
      double s = kIdct.c[x][0] * row[0];
      if (ac) for (int u = 1; u < 8; ++u) s += kIdct.c[x][u] * row[u];
      tmp[v * 8 + x] = s;
    }
  }
  for (int x = 0; x < 8; ++x)
    for (int y = 0; y < 8; ++y) {
      double s = 0;
      for (int v = 0; v < 8; ++v) s += kIdct.c[y][v] * tmp[v * 8 + x];
      long r = std::lround(s) + 128;
      out[y * stride + x] = uint8_t(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

inline uint16_t be16(const uint8_t* p) { return uint16_t((p[0] << 8) | p[1]); }

struct Decoder {
  const uint8_t* data; size_t len;
  std::string err;
  int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1;
  uint16_t qt[4][64] = {{0}};
  bool qt_present[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  Comp comp[3];
  int restart_interval = 0;
  bool adobe = false; int adobe_transform = 0;
  bool sof_seen = false, progressive = false;

  bool fail(const std::string& m) { err = m; return false; }

  static inline int16_t sat16(long long v) { return int16_t(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
  // DC predictor of a crafted stream: saturate far outside anything a real frame reaches (|DC| < 2^15) instead of
  // running into signed overflow after millions of maximal differences
  static inline int add_pred(int pred, int diff) { const int v = pred + diff; return v < -(1 << 24) ? -(1 << 24) : (v > (1 << 24) ? (1 << 24) : v); }

  // ---- one block of one scan.  Sequential frames: DC + all AC (ss = 0, se = 63, ah = al = 0).  Progressive frames
  // (ITU T.81 annex G): DC first / refinement, AC first / refinement with end-of-band runs shared across blocks.
  bool block_sequential(BitReader& br, Comp& cp, int16_t* blk) {
    const Huff& hd = dc[cp.td]; const Huff& ha = ac[cp.ta];
    int t = decode_sym(br, hd);
    if (t < 0 || t > 11) return fail("bad DC huffman code");
    cp.pred = add_pred(cp.pred, t ? extend(br.get(t), t) : 0);
    blk[0] = sat16(cp.pred);
    for (int k = 1; k < 64;) {
      int rs = decode_sym(br, ha);
      if (rs < 0) return fail("bad AC huffman code");
      int r = rs >> 4, sz = rs & 15;
      if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
      k += r;
      if (k > 63) return fail("AC index overflow");
      blk[kZigzag[k]] = sat16(extend(br.get(sz), sz));
      ++k;
    }
    return true;
  }
  bool block_dc_first(BitReader& br, Comp& cp, int16_t* blk, int al) {
    int t = decode_sym(br, dc[cp.td]);
    if (t < 0 || t > 11) return fail("bad DC huffman code");
    cp.pred = add_pred(cp.pred, t ? extend(br.get(t), t) : 0);
    blk[0] = sat16((long long)cp.pred * (1ll << al));
    return true;
  }
  static void block_dc_refine(BitReader& br, int16_t* blk, int al) {
    if (br.get(1)) blk[0] = int16_t(blk[0] | (1 << al));
  }
  bool block_ac_first(BitReader& br, const Huff& ha, int16_t* blk, int ss, int se, int al, int& eobrun) {
    if (eobrun > 0) { --eobrun; return true; }
    for (int k = ss; k <= se; ++k) {
      int rs = decode_sym(br, ha);
      if (rs < 0) return fail("bad AC huffman code");
      int r = rs >> 4, sz = rs & 15;
      if (sz) {
        k += r;
        if (k > 63) return fail("AC index overflow");
        blk[kZigzag[k]] = sat16(extend(br.get(sz), sz) * (1 << al));
      } else if (r == 15) {
        k += 15;
      } else {
        eobrun = (1 << r) - 1;
        if (r) eobrun += br.get(r);
        break;
      }
    }
    return true;
  }
  bool block_ac_refine(BitReader& br, const Huff& ha, int16_t* blk, int ss, int se, int al, int& eobrun) {
    const int p1 = 1 << al, m1 = -(1 << al);
    auto correct = [&](int16_t& c) { if (br.get(1) && (c & p1) == 0) c = sat16(c + (c >= 0 ? p1 : m1)); };
    int k = ss;
    if (eobrun == 0) {
      for (; k <= se; ++k) {
        int rs = decode_sym(br, ha);
        if (rs < 0) return fail("bad AC huffman code");
        int r = rs >> 4, sz = rs & 15, val = 0;
        if (sz) {
          if (sz != 1) return fail("bad AC refinement code");
          val = br.get(1) ? p1 : m1;
        } else if (r != 15) {
          eobrun = 1 << r;
          if (r) eobrun += br.get(r);
          break;
        }
        // pass over the coefficients that are already non-zero (each takes a correction bit) and r zero ones
        for (; k <= se; ++k) {
          int16_t& c = blk[kZigzag[k]];
          if (c != 0) correct(c);
          else if (--r < 0) break;
        }
        if (val) {
          if (k > se) return fail("AC refinement runs past the band");
          blk[kZigzag[k]] = int16_t(val);
        }
      }
    }
    if (eobrun > 0) {  // end of band: the remaining non-zero coefficients still take their correction bits
      for (; k <= se; ++k) {
        int16_t& c = blk[kZigzag[k]];
        if (c != 0) correct(c);
      }
      --eobrun;
    }
    return true;
  }

  // ---- one scan (SOS): `ns` components (indices into comp[]), spectral band ss..se, successive approximation ah / al
  bool decode_scan(const uint8_t* p, const uint8_t* end, const uint8_t** next, const int* ci, int ns, int ss, int se, int ah, int al) {
    BitReader br{p, end};
    int units_x, units_y;  // MCUs of an interleaved scan, or the blocks of the one component of a non-interleaved scan
    if (ns > 1) { units_x = (width + 8 * hmax - 1) / (8 * hmax); units_y = (height + 8 * vmax - 1) / (8 * vmax); }
    else { units_x = comp[ci[0]].nbx; units_y = comp[ci[0]].nby; }
    int rst_left = restart_interval, eobrun = 0;
    for (int i = 0; i < ns; ++i) comp[ci[i]].pred = 0;
    for (int uy = 0; uy < units_y; ++uy)
      for (int ux = 0; ux < units_x; ++ux) {
        if (restart_interval && rst_left == 0) {
          // byte-align, expect RSTn
          br.reset();
          const uint8_t* q = br.p;
          while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
          if (q + 1 >= end) return fail("missing restart marker");
          br.p = q + 2;
          for (int i = 0; i < ns; ++i) comp[ci[i]].pred = 0;
          eobrun = 0;
          rst_left = restart_interval;
        }
        for (int i = 0; i < ns; ++i) {
          Comp& cp = comp[ci[i]];
          const int nv = ns > 1 ? cp.v : 1, nh = ns > 1 ? cp.h : 1;
          for (int by = 0; by < nv; ++by)
            for (int bx = 0; bx < nh; ++bx) {
              const int gx = ns > 1 ? ux * cp.h + bx : ux, gy = ns > 1 ? uy * cp.v + by : uy;
              int16_t* blk = cp.coef.data() + (size_t(gy) * cp.bw + gx) * 64;
              bool ok = true;
              if (!progressive) ok = block_sequential(br, cp, blk);
              else if (ss == 0) { if (ah == 0) ok = block_dc_first(br, cp, blk, al); else block_dc_refine(br, blk, al); }
              else if (ah == 0) ok = block_ac_first(br, ac[cp.ta], blk, ss, se, al, eobrun);
              else ok = block_ac_refine(br, ac[cp.ta], blk, ss, se, al, eobrun);
              if (!ok) return false;
            }
        }
        if (restart_interval) --rst_left;
      }
    *next = br.p;
    return true;
  }

  // after the last scan: dequantise + inverse DCT of every block
  bool reconstruct() {
    for (int c = 0; c < ncomp; ++c) {
      Comp& cp = comp[c];
      if (!qt_present[cp.tq]) return fail("missing quantisation table");
      int qnat[64];
      for (int k = 0; k < 64; ++k) qnat[kZigzag[k]] = qt[cp.tq][k];
      cp.plane.assign(size_t(cp.bw) * 8 * cp.bh * 8, 0);
      int dq[64];
      for (int by = 0; by < cp.bh; ++by)
        for (int bx = 0; bx < cp.bw; ++bx) {
          const int16_t* blk = cp.coef.data() + (size_t(by) * cp.bw + bx) * 64;
          for (int k = 0; k < 64; ++k) dq[k] = int(blk[k]) * qnat[k];
          idct_block(dq, cp.plane.data() + size_t(by) * 8 * (cp.bw * 8) + size_t(bx) * 8, cp.bw * 8);
        }
      std::vector<int16_t>().swap(cp.coef);
    }
    return true;
  }

  bool parse() {
    if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG (no SOI)");
    const uint8_t* p = data + 2; const uint8_t* end = data + len;
    bool scanned = false;
    while (p + 4 <= end) {
      if (p[0] != 0xFF) { ++p; continue; }
      uint8_t m = p[1];
      if (m == 0xFF) { ++p; continue; }
      if (m == 0xD9) break;
      if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; }
      uint16_t L = be16(p + 2);
      if (L < 2 || p + 2 + L > end) return fail("truncated segment");
      const uint8_t* s = p + 4; const uint8_t* se = p + 2 + L;
      if (m == 0xDB) {
        while (s < se) {
          int pq = s[0] >> 4, tq = s[0] & 15; ++s;
          if (tq > 3) return fail("bad DQT id");
          for (int i = 0; i < 64; ++i) {
            if (pq) { if (s + 2 > se) return fail("bad DQT"); qt[tq][i] = be16(s); s += 2; }
            else { if (s + 1 > se) return fail("bad DQT"); qt[tq][i] = *s++; }
          }
          qt_present[tq] = true;
        }
      } else if (m == 0xC4) {
        while (s < se) {
          int tc = s[0] >> 4, th = s[0] & 15; ++s;
          if (th > 3 || tc > 1 || s + 16 > se) return fail("bad DHT");
          Huff& h = tc ? ac[th] : dc[th];
          int total = 0; h.bits[0] = 0;
          for (int i = 1; i <= 16; ++i) { h.bits[i] = s[i - 1]; total += s[i - 1]; }
          s += 16;
          if (total > 256 || s + total > se) return fail("bad DHT");
          std::memcpy(h.vals, s, total); s += total;
          if (!h.build()) return fail("bad DHT (over-subscribed code lengths)");
        }
      } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
        if (sof_seen) return fail("more than one frame header");
        progressive = m == 0xC2;
        if (L < 8) return fail("truncated SOF");
        if (s[0] != 8) return fail("only 8-bit JPEG supported");
        height = be16(s + 1); width = be16(s + 3); ncomp = s[5];
        if (!(ncomp == 1 || ncomp == 3) || width <= 0 || height <= 0) return fail("unsupported component count");
        if (L < 8 + 3 * ncomp) return fail("truncated SOF");
        hmax = vmax = 1;
        for (int c = 0; c < ncomp; ++c) {
          comp[c].id = s[6 + 3 * c]; comp[c].h = s[7 + 3 * c] >> 4; comp[c].v = s[7 + 3 * c] & 15;
          comp[c].tq = s[8 + 3 * c];
          if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4 || comp[c].tq > 3) return fail("bad SOF");
          if (comp[c].h > hmax) hmax = comp[c].h;
          if (comp[c].v > vmax) vmax = comp[c].v;
        }
        int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        for (int c = 0; c < ncomp; ++c) {
          comp[c].bw = mcux * comp[c].h; comp[c].bh = mcuy * comp[c].v;
          const int cw = (width * comp[c].h + hmax - 1) / hmax, ch = (height * comp[c].v + vmax - 1) / vmax;
          comp[c].nbx = (cw + 7) / 8; comp[c].nby = (ch + 7) / 8;
          // The coefficient store is 128 B per block.  A Huffman-coded block takes at least one bit, so a frame header
          // that claims more than 8 blocks per byte of the file cannot be backed by data: refuse it before allocating
          // (a 30-byte header claiming 65535 x 65535 would otherwise zero-fill 8 GiB per component).
          const size_t blocks = size_t(comp[c].bw) * comp[c].bh;
          if (blocks > (size_t(1) << 22)) return fail("image too large (more than 2^22 blocks per component)");
          if (blocks > 8 * len + 64) return fail("frame header claims more blocks than the file could hold");
          comp[c].coef.assign(blocks * 64, 0);
        }
        sof_seen = true;
      } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
        char what[192];
        std::snprintf(what, sizeof what, "JPEG frame type SOF%d (%s) not supported: Huffman baseline / extended sequential / progressive only",
                      m - 0xC0, m >= 0xC9 ? "arithmetic coding" : "lossless / differential");
        return fail(what);
      } else if (m == 0xDD) {
        if (L < 4) return fail("truncated DRI");
        restart_interval = be16(s);
      } else if (m == 0xEE && L >= 14 && std::memcmp(s, "Adobe", 5) == 0) {
        adobe = true; adobe_transform = s[11];
      } else if (m == 0xDA) {
        if (!sof_seen) return fail("SOS before SOF");
        if (L < 3) return fail("truncated SOS");
        int ns = s[0];
        if (ns < 1 || ns > ncomp) return fail("bad SOS component count");
        if (L < 6 + 2 * ns) return fail("truncated SOS");
        int ci[3];
        for (int i = 0; i < ns; ++i) {
          int cid = s[1 + 2 * i], tbl = s[2 + 2 * i];
          int c = -1;
          for (int k = 0; k < ncomp; ++k) if (comp[k].id == cid) c = k;
          if (c < 0) return fail("bad SOS component");
          for (int j = 0; j < i; ++j) if (ci[j] == c) return fail("SOS names a component twice");
          ci[i] = c;
          comp[c].td = tbl >> 4; comp[c].ta = tbl & 15;
          if (comp[c].td > 3 || comp[c].ta > 3) return fail("bad SOS table id");
        }
        const int ss = s[1 + 2 * ns], sp_end = s[2 + 2 * ns], ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
        if (!progressive) {
          if (ss != 0 || sp_end != 63 || ah != 0 || al != 0) return fail("bad SOS parameters for a sequential frame");
        } else {
          if (ss > sp_end || sp_end > 63 || al > 13 || ah > 13 || (ss == 0 && sp_end != 0) || (ss > 0 && ns != 1) || (ah != 0 && ah != al + 1))
            return fail("bad SOS parameters for a progressive frame");
        }
        for (int i = 0; i < ns; ++i) {  // the tables this scan decodes with must have been defined
          const Comp& cp = comp[ci[i]];
          const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss > 0;
          if ((need_dc && !dc[cp.td].present) || (need_ac && !ac[cp.ta].present)) return fail("scan references a missing table");
        }
        if (ns > 1) {  // an interleaved scan must fit T.81's limit of 10 blocks per MCU
          int blocks = 0;
          for (int i = 0; i < ns; ++i) blocks += comp[ci[i]].h * comp[ci[i]].v;
          if (blocks > 10) return fail("too many blocks per MCU");
        }
        const uint8_t* next = nullptr;
        if (!decode_scan(se, end, &next, ci, ns, ss, sp_end, ah, al)) return false;
        scanned = true;
        p = next;
        continue;
      }
      p += 2 + L;
    }
    if (!scanned) return fail("no image data");
    return reconstruct();
  }


  // libjpeg-style "fancy" (triangle filter) upsampling of one component to full size
  void upsample(const Comp& c, std::vector<uint8_t>& out) const {
    const int W = width, H = height, pw = c.bw * 8, ph = c.bh * 8;
    out.resize(size_t(W) * H);
    const int fx = hmax / c.h, fy = vmax / c.v;
    const bool exact = (hmax % c.h == 0) && (vmax % c.v == 0);
    // number of valid source samples (rest of the plane is MCU padding)
    const int sw = (W * c.h + hmax - 1) / hmax, sh = (H * c.v + vmax - 1) / vmax;
    auto S = [&](int x, int y) -> int {
      x = x < 0 ? 0 : (x >= sw ? sw - 1 : x);
      y = y < 0 ? 0 : (y >= sh ? sh - 1 : y);
      (void)ph;
      return c.plane[size_t(y) * pw + x];
    };
    if (exact && fx == 1 && fy == 1) {
      for (int y = 0; y < H; ++y) std::memcpy(&out[size_t(y) * W], &c.plane[size_t(y) * pw], W);
    } else if (exact && fx == 2 && fy == 2) {
      for (int y = 0; y < H; ++y) {
        int sy = y >> 1, ny = (y & 1) ? sy + 1 : sy - 1;
        for (int x = 0; x < W; ++x) {
          int sx = x >> 1, nx = (x & 1) ? sx + 1 : sx - 1;
          int thiscol = 3 * S(sx, sy) + S(sx, ny), nextcol = 3 * S(nx, sy) + S(nx, ny);
          out[size_t(y) * W + x] = uint8_t((3 * thiscol + nextcol + ((x & 1) ? 7 : 8)) >> 4);
        }
      }
    } else if (exact && fx == 2 && fy == 1) {
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          int sx = x >> 1, nx = (x & 1) ? sx + 1 : sx - 1;
          out[size_t(y) * W + x] = uint8_t((3 * S(sx, y) + S(nx, y) + ((x & 1) ? 2 : 1)) >> 2);
        }
    } else if (exact && fx == 1 && fy == 2) {
      for (int y = 0; y < H; ++y) {
        int sy = y >> 1, ny = (y & 1) ? sy + 1 : sy - 1;
        for (int x = 0; x < W; ++x)
          out[size_t(y) * W + x] = uint8_t((3 * S(x, sy) + S(x, ny) + ((y & 1) ? 2 : 1)) >> 2);
      }
    } else {  // uncommon ratios: nearest
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) out[size_t(y) * W + x] = uint8_t(S(x * c.h / hmax, y * c.v / vmax));
    }
  }

  void to_rgb(uint8_t* rgb) const {
    const size_t n = size_t(width) * height;
    std::vector<uint8_t> p0, p1, p2;
    upsample(comp[0], p0);
    if (ncomp == 1) {
      for (size_t i = 0; i < n; ++i) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = p0[i];
      return;
    }
    upsample(comp[1], p1);
    upsample(comp[2], p2);
    const bool ycc = adobe ? adobe_transform != 0 : true;
    auto clamp8 = [](int v) { return uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (size_t i = 0; i < n; ++i) {
      if (!ycc) { rgb[3 * i] = p0[i]; rgb[3 * i + 1] = p1[i]; rgb[3 * i + 2] = p2[i]; continue; }
      // JFIF YCbCr -> RGB, 16-bit fixed point like libjpeg's jdcolor.c
      int y = p0[i], cb = p1[i] - 128, cr = p2[i] - 128;
      int r = y + ((91881 * cr + 32768) >> 16);
      int g = y + ((-22554 * cb - 46802 * cr + 32768) >> 16);
      int b = y + ((116130 * cb + 32768) >> 16);
      rgb[3 * i] = clamp8(r); rgb[3 * i + 1] = clamp8(g); rgb[3 * i + 2] = clamp8(b);
    }
  }
};

thread_local std::string g_jpeg_err;

}  // namespace

extern "C" const char* rt_jpeg_last_error(void) { return g_jpeg_err.c_str(); }

extern "C" int rt_jpeg_decode_mem(const uint8_t* data, size_t len, uint8_t** rgb8, uint32_t* w, uint32_t* h) {
  if (!data || !rgb8 || !w || !h) return RT_ERR_INVALID;
  // nothing may leave an extern "C" function by exception: an allocation failure inside the decoder is a texture error
  uint8_t* out = nullptr;
  try {
    Decoder d; d.data = data; d.len = len;
    if (!d.parse()) { g_jpeg_err = d.err; return RT_ERR_TEXTURE; }
    out = static_cast<uint8_t*>(std::malloc(size_t(d.width) * d.height * 3));
    if (!out) { g_jpeg_err = "out of memory"; return RT_ERR_TEXTURE; }
    d.to_rgb(out);
    *rgb8 = out; *w = uint32_t(d.width); *h = uint32_t(d.height);
    return RT_OK;
  } catch (const std::bad_alloc&) {
    std::free(out);
    g_jpeg_err = "out of memory while decoding";
    return RT_ERR_TEXTURE;
  } catch (...) {
    std::free(out);
    g_jpeg_err = "internal decoder error";
    return RT_ERR_TEXTURE;
  }
}

extern "C" int rt_jpeg_decode_file(const char* path, uint8_t** rgb8, uint32_t* w, uint32_t* h) {
  if (!path) return RT_ERR_INVALID;
  FILE* f = std::fopen(path, "rb");
  if (!f) { g_jpeg_err = std::string("cannot open ") + path; return RT_ERR_TEXTURE; }
  std::vector<uint8_t> buf;
  try {
    uint8_t tmp[65536]; size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  } catch (const std::bad_alloc&) {
    std::fclose(f);
    g_jpeg_err = std::string("out of memory reading ") + path;
    return RT_ERR_TEXTURE;
  }
  std::fclose(f);
  return rt_jpeg_decode_mem(buf.data(), buf.size(), rgb8, w, h);
}
