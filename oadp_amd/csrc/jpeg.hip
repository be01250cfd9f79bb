// jpeg.hip — baseline JPEG -> uint8 RGB on the device, bit-identical to Pillow / libjpeg-turbo
// (the step in front of the crop pipeline; the reference decodes with
// PIL.Image.open(...).convert('RGB') in its DataLoader workers, oadp/oake/base.py:53).
//
// Split of work: the entropy-coded segment is a serial bit stream, so the Huffman decode stays on
// the host (one pass, 9-bit look-ahead tables, jdhuff.c's scheme) and produces the quantised DCT
// coefficients; everything arithmetic runs on the GPU:
//   jpeg_idct_kernel    dequantise + jpeg_idct_islow (jidctint.c, 13-bit fixed point, two passes)
//   jpeg_rgb_kernel     "fancy" triangle-filter chroma upsampling (jdsample.c h2v1 / h2v2 / h1v2)
//                       fused with the fixed-point YCbCr -> RGB of jdcolor.c, HWC uint8 out
// Scope: 8-bit Huffman-coded DCT frames — baseline / extended sequential (SOF0/SOF1, one interleaved
// scan) and progressive (SOF2, any scan script) — 1 or 3 components, sampling factors 1 or 2, restart
// intervals.  Anything else returns JPEG_UNSUPPORTED (the caller decides;
// there is no silent CPU path).  The CPU restatement these kernels are tested against is
// oracle/jpeg_ref.py, itself pinned bit-exactly to the Pillow in the image.
#include <string.h>

#include <string>

#include "common.h"
#include "kernels.h"

namespace oake {

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int kLook = 10;  // look-ahead bits

struct Huff {
  // jdhuff.c derived table: codes of <= kLook bits resolve in one lookup, longer ones by length
  uint16_t look[1 << kLook];  // (length << 8) | symbol, 0 = code longer than kLook bits
  // AC shortcut: when code + magnitude bits fit the look-ahead, one lookup yields everything:
  // (coefficient << 16) | (run << 8) | total bits consumed; 0 = take the slow path
  int32_t fast_ac[1 << kLook];
  int32_t maxcode[18];
  int32_t valoff[17];
  uint8_t vals[256];
  bool present = false;

  // false: the code lengths oversubscribe the code space (corrupt DHT)
  bool build(const uint8_t* bits, const uint8_t* v, int n) {
    memset(look, 0, sizeof(look));
    memset(fast_ac, 0, sizeof(fast_ac));
    memcpy(vals, v, n);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valoff[len] = k - code;
      for (int i = 0; i < bits[len - 1]; ++i, ++k, ++code) {
        if (len <= kLook) {
          const int base = code << (kLook - len);
          for (int f = 0; f < (1 << (kLook - len)); ++f) look[base + f] = (uint16_t)((len << 8) | v[k]);
        }
      }
      if (code > (1 << len)) return false;
      maxcode[len] = bits[len - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    for (int w = 0; w < (1 << kLook); ++w) {
      const int e = look[w];
      if (!e) continue;
      const int len = e >> 8, rs = e & 0xFF, run = rs >> 4, mag = rs & 15;
      if (mag == 0 || len + mag > kLook) continue;
      int val = (w >> (kLook - len - mag)) & ((1 << mag) - 1);
      if (val < (1 << (mag - 1))) val += 1 - (1 << mag);
      fast_ac[w] = (int32_t)((uint32_t)val << 16) | (run << 8) | (len + mag);
    }
    present = true;
    return true;
  }
};

struct BitReader {
  const uint8_t* d;
  size_t n, pos;
  uint64_t acc = 0;
  int cnt = 0;
  bool hit_marker = false;

  inline void fill() {
    while (cnt <= 56) {
      uint32_t b = 0;
      if (!hit_marker && pos < n) {
        b = d[pos];
        if (b == 0xFF) {
          const uint8_t nx = pos + 1 < n ? d[pos + 1] : 0xD9;
          if (nx == 0) {
            pos += 2;
          } else {
            hit_marker = true;  // feed zeros from here on, as libjpeg does
            b = 0;
          }
        } else {
          ++pos;
        }
      }
      acc = (acc << 8) | b;
      cnt += 8;
    }
  }
  inline uint32_t peek(int k) { return (uint32_t)(acc >> (cnt - k)) & ((1u << k) - 1); }
  inline void drop(int k) { cnt -= k; }
  inline int decode(const Huff& t) {
    if (cnt < 16) fill();
    const uint16_t e = t.look[peek(kLook)];
    if (e) {
      drop(e >> 8);
      return e & 0xFF;
    }
    int len = kLook + 1;
    int32_t code = (int32_t)peek(kLook + 1);
    while (len <= 16 && code > t.maxcode[len]) {
      ++len;
      code = (int32_t)peek(len);
    }
    if (len > 16) return -1;
    drop(len);
    return t.vals[(code + t.valoff[len]) & 255];
  }
  inline int receive_extend(int s) {
    if (cnt < s) fill();
    const int v = (int)peek(s);
    drop(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
  }
  // byte-align and step over the expected RSTn marker
  bool restart() {
    cnt = 0;
    acc = 0;
    hit_marker = false;
    while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) ++pos;
    if (pos + 1 >= n) return false;
    pos += 2;
    return true;
  }
};

struct Tables {
  Huff dc[4], ac[4];
  int td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
  int restart_interval = 0;
  size_t scan_pos = 0;
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// Walks the marker segments up to and including SOS.  frame: always filled; tables: optional.
int parse_markers(const uint8_t* d, size_t n, JpegFrame* f, Tables* t, std::string* err) {
  auto fail = [&](int rc, const char* m) {
    if (err) *err = m;
    return rc;
  };
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail(JPEG_INVALID, "not a JPEG (no SOI)");
  memset(f, 0, sizeof(*f));
  uint16_t qt[4][64];
  bool have_q[4] = {false, false, false, false};
  int comp_id[3] = {0, 0, 0}, comp_tq[3] = {0, 0, 0};
  bool have_sof = false;
  size_t pos = 2;
  for (;;) {
    while (pos < n && d[pos] != 0xFF) ++pos;
    while (pos < n && d[pos] == 0xFF) ++pos;
    if (pos >= n) return fail(JPEG_INVALID, "truncated before SOS");
    const uint8_t m = d[pos++];
    if (m == 0xD9) return fail(JPEG_INVALID, "EOI before SOS");
    if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (pos + 2 > n) return fail(JPEG_INVALID, "truncated segment");
    const int seg = be16(d + pos);
    if (seg < 2 || pos + seg > n) return fail(JPEG_INVALID, "bad segment length");
    const uint8_t* b = d + pos + 2;
    const int len = seg - 2;
    if (m == 0xDB) {
      int i = 0;
      while (i < len) {
        const int pq = b[i] >> 4, tq = b[i] & 15;
        ++i;
        if (tq > 3 || i + (pq ? 128 : 64) > len) return fail(JPEG_INVALID, "bad DQT");
        for (int k = 0; k < 64; ++k) {
          qt[tq][kZigzag[k]] = pq ? (uint16_t)be16(b + i + 2 * k) : b[i + k];
        }
        i += pq ? 128 : 64;
        have_q[tq] = true;
      }
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      if (len < 6 || b[0] != 8) return fail(JPEG_UNSUPPORTED, "only 8-bit samples");
      f->progressive = m == 0xC2;
      f->height = be16(b + 1);
      f->width = be16(b + 3);
      f->ncomp = b[5];
      if (f->ncomp != 1 && f->ncomp != 3) return fail(JPEG_UNSUPPORTED, "only 1- or 3-component JPEG");
      if (len < 6 + 3 * f->ncomp) return fail(JPEG_INVALID, "bad SOF");
      for (int c = 0; c < f->ncomp; ++c) {
        comp_id[c] = b[6 + 3 * c];
        f->h[c] = b[7 + 3 * c] >> 4;
        f->v[c] = b[7 + 3 * c] & 15;
        comp_tq[c] = b[8 + 3 * c];
        if (comp_tq[c] > 3) return fail(JPEG_INVALID, "bad quantisation table index");
      }
      have_sof = true;
    } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xC7) || (m >= 0xC9 && m <= 0xCB) ||
               (m >= 0xCD && m <= 0xCF)) {
      return fail(JPEG_UNSUPPORTED, "only Huffman-coded sequential / progressive DCT JPEG (arithmetic, lossless, hierarchical not supported)");
    } else if (m == 0xC4) {
      int i = 0;
      while (i < len) {
        if (i + 17 > len) return fail(JPEG_INVALID, "bad DHT");
        const int tc = b[i] >> 4, th = b[i] & 15;
        int cnt = 0;
        for (int k = 0; k < 16; ++k) cnt += b[i + 1 + k];
        if (th > 3 || tc > 1 || cnt > 256 || i + 17 + cnt > len) return fail(JPEG_INVALID, "bad DHT");
        if (t && !(tc ? t->ac[th] : t->dc[th]).build(b + i + 1, b + i + 17, cnt))
          return fail(JPEG_INVALID, "bad DHT (oversubscribed code lengths)");
        i += 17 + cnt;
      }
    } else if (m == 0xDD) {
      if (len < 2) return fail(JPEG_INVALID, "bad DRI");
      if (t) t->restart_interval = be16(b);
    } else if (m == 0xDA) {
      if (!have_sof) return fail(JPEG_INVALID, "SOS before SOF");
      if (f->progressive) break;  // scans are walked by decode_progressive
      if (len < 1 || b[0] != f->ncomp || len < 1 + 2 * f->ncomp)
        return fail(JPEG_UNSUPPORTED, "only single interleaved scans");
      for (int k = 0; k < f->ncomp; ++k) {
        int c = -1;
        for (int j = 0; j < f->ncomp; ++j)
          if (comp_id[j] == b[1 + 2 * k]) c = j;
        if (c != k) return fail(JPEG_UNSUPPORTED, "scan component order differs from frame order");
        if (t) {
          t->td[c] = b[2 + 2 * k] >> 4;
          t->ta[c] = b[2 + 2 * k] & 15;
          if (t->td[c] > 3 || t->ta[c] > 3) return fail(JPEG_INVALID, "bad Huffman table index");
        }
      }
      if (t) t->scan_pos = pos + seg;
      break;
    }
    pos += seg;
  }
  for (int c = 0; c < 3; ++c) f->comp_id[c] = comp_id[c];
  if (f->width <= 0 || f->height <= 0) return fail(JPEG_INVALID, "empty image");
  if (f->ncomp == 1) f->h[0] = f->v[0] = 1;  // a single-component scan is not interleaved
  f->hmax = f->vmax = 1;
  for (int c = 0; c < f->ncomp; ++c) {
    if (f->h[c] < 1 || f->h[c] > 2 || f->v[c] < 1 || f->v[c] > 2)
      return fail(JPEG_UNSUPPORTED, "sampling factors other than 1 and 2");
    f->hmax = f->h[c] > f->hmax ? f->h[c] : f->hmax;
    f->vmax = f->v[c] > f->vmax ? f->v[c] : f->vmax;
    if (!have_q[comp_tq[c]]) return fail(JPEG_INVALID, "missing quantisation table");
    memcpy(f->q[c], qt[comp_tq[c]], sizeof(f->q[c]));
  }
  if (f->ncomp == 3 && (f->h[0] != f->hmax || f->v[0] != f->vmax || f->h[1] != f->h[2] || f->v[1] != f->v[2]))
    return fail(JPEG_UNSUPPORTED, "luma must carry the maximum sampling factors, chroma planes must match");
  f->mcux = (f->width + 8 * f->hmax - 1) / (8 * f->hmax);
  f->mcuy = (f->height + 8 * f->vmax - 1) / (8 * f->vmax);
  long co = 0, po = 0;
  for (int c = 0; c < f->ncomp; ++c) {
    f->bx[c] = f->mcux * f->h[c];
    f->by[c] = f->mcuy * f->v[c];
    f->coef_off[c] = co;
    f->plane_off[c] = po;
    co += (long)f->bx[c] * f->by[c] * 64;
    po += (long)f->bx[c] * f->by[c] * 64;
  }
  f->total_coefs = co;
  f->total_plane_bytes = po;
  return JPEG_OK;
}

// ---- progressive frames (ITU T.81 Annex G, jdphuff.c) ------------------------------------------
// One scan: DC first / DC refinement (possibly interleaved) or AC first / AC refinement (one
// component, over that component's own blocks, not the MCU-padded ones).
int decode_progressive_scan(BitReader& br, const JpegFrame& fr, const Tables& t, const int* scan_ci, int ns,
                            int ss, int se, int ah, int al, int16_t* coefs, std::string* err) {
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return JPEG_INVALID;
  };
  if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) return fail("bad progressive scan header");
  int pred[3] = {0, 0, 0};
  int eobrun = 0;
  int ux_n, uy_n;
  if (ns > 1) {
    ux_n = fr.mcux;
    uy_n = fr.mcuy;
  } else {
    const int c = scan_ci[0];
    ux_n = ((fr.width * fr.h[c] + fr.hmax - 1) / fr.hmax + 7) / 8;
    uy_n = ((fr.height * fr.v[c] + fr.vmax - 1) / fr.vmax + 7) / 8;
  }
  const int p1 = 1 << al, m1 = -(1 << al);
  long count = 0;
  for (int uy = 0; uy < uy_n; ++uy) {
    for (int ux = 0; ux < ux_n; ++ux) {
      if (t.restart_interval && count && count % t.restart_interval == 0) {
        if (!br.restart()) return fail("missing restart marker");
        pred[0] = pred[1] = pred[2] = 0;
        eobrun = 0;
      }
      ++count;
      for (int sc = 0; sc < ns; ++sc) {
        const int c = scan_ci[sc];
        const int nby = ns > 1 ? fr.v[c] : 1, nbx = ns > 1 ? fr.h[c] : 1;
        for (int by = 0; by < nby; ++by) {
          for (int bx = 0; bx < nbx; ++bx) {
            const long brow = ns > 1 ? (long)uy * fr.v[c] + by : uy;
            const long bcol = ns > 1 ? (long)ux * fr.h[c] + bx : ux;
            int16_t* blk = coefs + fr.coef_off[c] + (brow * fr.bx[c] + bcol) * 64;
            if (ss == 0) {
              if (ah == 0) {
                const int s = br.decode(t.dc[t.td[c]]);
                if (s < 0 || s > 15) return fail("corrupt DC code");
                if (s) pred[c] += br.receive_extend(s);
                blk[0] = (int16_t)(pred[c] * (1 << al));
              } else {
                if (br.cnt < 1) br.fill();
                if (br.peek(1)) blk[0] = (int16_t)(blk[0] | p1);
                br.drop(1);
              }
              continue;
            }
            const Huff& ha = t.ac[t.ta[c]];
            if (ah == 0) {  // AC first
              if (eobrun) {
                --eobrun;
                continue;
              }
              for (int k = ss; k <= se;) {
                const int rs = br.decode(ha);
                if (rs < 0) return fail("corrupt AC code");
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) {
                  if (r == 15) {
                    k += 16;
                    continue;
                  }
                  eobrun = (1 << r) - 1;
                  if (r) {
                    if (br.cnt < r) br.fill();
                    eobrun += (int)br.peek(r);
                    br.drop(r);
                  }
                  break;
                }
                k += r;
                if (k > se) return fail("AC run past the band");
                blk[kZigzag[k]] = (int16_t)(br.receive_extend(s) * (1 << al));
                ++k;
              }
              continue;
            }
            // AC refinement
            int k = ss;
            if (eobrun == 0) {
              while (k <= se) {
                const int rs = br.decode(ha);
                if (rs < 0) return fail("corrupt AC code");
                int r = rs >> 4;
                const int s = rs & 15;
                int val = 0;
                if (s) {
                  if (s != 1) return fail("corrupt AC refinement");
                  if (br.cnt < 1) br.fill();
                  val = br.peek(1) ? p1 : m1;
                  br.drop(1);
                } else if (r != 15) {
                  eobrun = 1 << r;
                  if (r) {
                    if (br.cnt < r) br.fill();
                    eobrun += (int)br.peek(r);
                    br.drop(r);
                  }
                  break;
                }
                while (k <= se) {
                  int16_t& cz = blk[kZigzag[k]];
                  if (cz != 0) {
                    if (br.cnt < 1) br.fill();
                    if (br.peek(1) && (cz & p1) == 0) cz = (int16_t)(cz + (cz >= 0 ? p1 : m1));
                    br.drop(1);
                  } else {
                    if (r == 0) {
                      if (val) cz = (int16_t)val;
                      ++k;
                      break;
                    }
                    --r;
                  }
                  ++k;
                }
              }
            }
            if (eobrun > 0) {
              for (; k <= se; ++k) {
                int16_t& cz = blk[kZigzag[k]];
                if (cz != 0) {
                  if (br.cnt < 1) br.fill();
                  if (br.peek(1) && (cz & p1) == 0) cz = (int16_t)(cz + (cz >= 0 ? p1 : m1));
                  br.drop(1);
                }
              }
              --eobrun;
            }
          }
        }
      }
    }
  }
  return JPEG_OK;
}

// All scans of a progressive file: tables may change between scans.
int decode_progressive(const uint8_t* d, size_t n, const JpegFrame& fr, int16_t* coefs, std::string* err) {
  auto fail = [&](int rc, const char* m) {
    if (err) *err = m;
    return rc;
  };
  Tables t;
  size_t pos = 2;
  int scans = 0;
  for (;;) {
    while (pos < n && d[pos] != 0xFF) ++pos;
    while (pos < n && d[pos] == 0xFF) ++pos;
    if (pos >= n) break;
    const uint8_t m = d[pos++];
    if (m == 0xD9) break;
    if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (pos + 2 > n) return fail(JPEG_INVALID, "truncated segment");
    const int seg = be16(d + pos);
    if (seg < 2 || pos + seg > n) return fail(JPEG_INVALID, "bad segment length");
    const uint8_t* b = d + pos + 2;
    const int len = seg - 2;
    if (m == 0xC4) {
      int i = 0;
      while (i < len) {
        if (i + 17 > len) return fail(JPEG_INVALID, "bad DHT");
        const int tc = b[i] >> 4, th = b[i] & 15;
        int cnt = 0;
        for (int k = 0; k < 16; ++k) cnt += b[i + 1 + k];
        if (th > 3 || tc > 1 || cnt > 256 || i + 17 + cnt > len) return fail(JPEG_INVALID, "bad DHT");
        if (!(tc ? t.ac[th] : t.dc[th]).build(b + i + 1, b + i + 17, cnt))
          return fail(JPEG_INVALID, "bad DHT (oversubscribed code lengths)");
        i += 17 + cnt;
      }
    } else if (m == 0xDD) {
      if (len < 2) return fail(JPEG_INVALID, "bad DRI");
      t.restart_interval = be16(b);
    } else if (m == 0xDA) {
      if (len < 1) return fail(JPEG_INVALID, "bad SOS");
      const int ns = b[0];
      if (ns < 1 || ns > fr.ncomp || len < 4 + 2 * ns) return fail(JPEG_INVALID, "bad SOS");
      int scan_ci[3];
      for (int k = 0; k < ns; ++k) {
        int c = -1;
        for (int j = 0; j < fr.ncomp; ++j)
          if (fr.comp_id[j] == b[1 + 2 * k]) c = j;
        if (c < 0) return fail(JPEG_INVALID, "scan names an unknown component");
        scan_ci[k] = c;
        t.td[c] = b[2 + 2 * k] >> 4;
        t.ta[c] = b[2 + 2 * k] & 15;
        if (t.td[c] > 3 || t.ta[c] > 3) return fail(JPEG_INVALID, "bad Huffman table index");
      }
      const int ss = b[1 + 2 * ns], se = b[2 + 2 * ns], ah = b[3 + 2 * ns] >> 4, al = b[3 + 2 * ns] & 15;
      for (int k = 0; k < ns; ++k) {
        const int c = scan_ci[k];
        if (ss == 0 && ah == 0 && !t.dc[t.td[c]].present) return fail(JPEG_INVALID, "missing Huffman table");
        if (ss > 0 && !t.ac[t.ta[c]].present) return fail(JPEG_INVALID, "missing Huffman table");
      }
      BitReader br{d, n, pos + (size_t)seg};
      const int rc = decode_progressive_scan(br, fr, t, scan_ci, ns, ss, se, ah, al, coefs, err);
      if (rc != JPEG_OK) return rc;
      ++scans;
      pos = br.pos;  // the reader stops in front of the next marker
      continue;
    }
    pos += seg;
  }
  if (!scans) return fail(JPEG_INVALID, "no scan found");
  return JPEG_OK;
}

}  // namespace

int jpeg_read_frame(const uint8_t* data, size_t n, JpegFrame* frame, std::string* err) {
  return parse_markers(data, n, frame, nullptr, err);
}

int jpeg_decode_coefs(const uint8_t* data, size_t n, const JpegFrame& fr, int16_t* coefs,
                      std::string* err) {
  memset(coefs, 0, (size_t)fr.total_coefs * sizeof(int16_t));
  if (fr.progressive) return decode_progressive(data, n, fr, coefs, err);
  JpegFrame f2;
  Tables t;
  int rc = parse_markers(data, n, &f2, &t, err);
  if (rc != JPEG_OK) return rc;
  for (int c = 0; c < fr.ncomp; ++c)
    if (!t.dc[t.td[c]].present || !t.ac[t.ta[c]].present) {
      if (err) *err = "missing Huffman table";
      return JPEG_INVALID;
    }
  BitReader br{data, n, t.scan_pos};
  int pred[3] = {0, 0, 0};
  long count = 0;
  for (int my = 0; my < fr.mcuy; ++my) {
    for (int mx = 0; mx < fr.mcux; ++mx) {
      if (t.restart_interval && count && count % t.restart_interval == 0) {
        if (!br.restart()) {
          if (err) *err = "missing restart marker";
          return JPEG_INVALID;
        }
        pred[0] = pred[1] = pred[2] = 0;
      }
      ++count;
      for (int c = 0; c < fr.ncomp; ++c) {
        const Huff& hd = t.dc[t.td[c]];
        const Huff& ha = t.ac[t.ta[c]];
        for (int by = 0; by < fr.v[c]; ++by) {
          for (int bx = 0; bx < fr.h[c]; ++bx) {
            int16_t* blk = coefs + fr.coef_off[c] +
                           ((long)(my * fr.v[c] + by) * fr.bx[c] + (mx * fr.h[c] + bx)) * 64;
            int s = br.decode(hd);
            if (s < 0 || s > 15) {
              if (err) *err = "corrupt DC code";
              return JPEG_INVALID;
            }
            if (s) pred[c] += br.receive_extend(s);
            blk[0] = (int16_t)pred[c];
            for (int k = 1; k < 64;) {
              if (br.cnt < 32) br.fill();
              const int32_t fa = ha.fast_ac[br.peek(kLook)];
              if (fa) {  // run, magnitude bits and value in one lookup
                k += (fa >> 8) & 15;
                if (k > 63) {
                  if (err) *err = "AC run past the block";
                  return JPEG_INVALID;
                }
                br.drop(fa & 255);
                blk[kZigzag[k]] = (int16_t)(fa >> 16);
                ++k;
                continue;
              }
              const int rs = br.decode(ha);
              if (rs < 0) {
                if (err) *err = "corrupt AC code";
                return JPEG_INVALID;
              }
              const int r = rs >> 4;
              s = rs & 15;
              if (s == 0) {
                if (r == 15) {
                  k += 16;
                  continue;
                }
                break;
              }
              k += r;
              if (k > 63) {
                if (err) *err = "AC run past the block";
                return JPEG_INVALID;
              }
              blk[kZigzag[k]] = (int16_t)br.receive_extend(s);
              ++k;
            }
          }
        }
      }
    }
  }
  return JPEG_OK;
}

// ---- device side ----------------------------------------------------------------------------
namespace {

struct FrameDev {  // the part of JpegFrame the kernels need, by value
  int width, height, ncomp, hmax, vmax;
  int h[3], v[3], bx[3], by[3];
  long coef_off[3], plane_off[3];
  long nblocks[3];
  uint16_t q[3][64];  // natural order
};

constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270;
constexpr int F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299, F_1_847 = 15137;
constexpr int F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 8-point pass of jpeg_idct_islow (jidctint.c), in place on d[0..7]
__device__ __forceinline__ void idct8(int (&d)[8], int shift) {
  int z2 = d[2], z3 = d[6];
  int z1 = (z2 + z3) * F_0_541;
  int tmp2 = z1 + z3 * (-F_1_847);
  int tmp3 = z1 + z2 * F_0_765;
  z2 = d[0];
  z3 = d[4];
  int tmp0 = (z2 + z3) << CONST_BITS;
  int tmp1 = (z2 - z3) << CONST_BITS;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = d[7];
  tmp1 = d[5];
  tmp2 = d[3];
  tmp3 = d[1];
  z1 = tmp0 + tmp3;
  z2 = tmp1 + tmp2;
  z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * F_1_175;
  tmp0 *= F_0_298;
  tmp1 *= F_2_053;
  tmp2 *= F_3_072;
  tmp3 *= F_1_501;
  z1 *= -F_0_899;
  z2 *= -F_2_562;
  z3 *= -F_1_961;
  z4 *= -F_0_390;
  z3 += z5;
  z4 += z5;
  tmp0 += z1 + z3;
  tmp1 += z2 + z4;
  tmp2 += z2 + z3;
  tmp3 += z1 + z4;
  d[0] = descale(tmp10 + tmp3, shift);
  d[7] = descale(tmp10 - tmp3, shift);
  d[1] = descale(tmp11 + tmp2, shift);
  d[6] = descale(tmp11 - tmp2, shift);
  d[2] = descale(tmp12 + tmp1, shift);
  d[5] = descale(tmp12 - tmp1, shift);
  d[3] = descale(tmp13 + tmp0, shift);
  d[4] = descale(tmp13 - tmp0, shift);
}

// jdmaster.c prepare_range_limit_table, post-IDCT half, indexed with (x & RANGE_MASK)
__device__ __forceinline__ uint8_t range_limit(int x) {
  const int t = x & 1023;
  return (uint8_t)(t < 128 ? t + 128 : (t < 512 ? 255 : (t < 896 ? 0 : t - 896)));
}

// 8 threads per 8x8 block: thread j owns column j in pass 1 and row j in pass 2; the 8x8 workspace
// is exchanged through LDS ([block][8][9] ints, the pad keeps both access directions conflict-free).
__global__ __launch_bounds__(256) void jpeg_idct_kernel(FrameDev f, const int16_t* __restrict__ coefs,
                                                        uint8_t* __restrict__ planes, long total_blocks) {
  __shared__ int ws[32][8][9];
  const int j = threadIdx.x & 7;
  const int lb = threadIdx.x >> 3;
  const long blk = (long)blockIdx.x * 32 + lb;
  const bool live = blk < total_blocks;
  int c = 0;
  long b = blk;
  if (live) {
    while (c + 1 < f.ncomp && b >= f.nblocks[c]) {
      b -= f.nblocks[c];
      ++c;
    }
  }
  int d[8];
  if (live) {
    const int16_t* src = coefs + f.coef_off[c] + b * 64;
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = (int)src[r * 8 + j] * (int)f.q[c][r * 8 + j];
    idct8(d, CONST_BITS - PASS1_BITS);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[lb][r][j] = d[r];
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = ws[lb][j][k];
    idct8(d, CONST_BITS + PASS1_BITS + 3);
    const int brow = (int)(b / f.bx[c]), bcol = (int)(b - (long)brow * f.bx[c]);
    uint8_t* dst = planes + f.plane_off[c] + ((long)(brow * 8 + j) * f.bx[c] + bcol) * 8;
    uint2 o;
    o.x = range_limit(d[0]) | (range_limit(d[1]) << 8) | (range_limit(d[2]) << 16) | ((uint32_t)range_limit(d[3]) << 24);
    o.y = range_limit(d[4]) | (range_limit(d[5]) << 8) | (range_limit(d[6]) << 16) | ((uint32_t)range_limit(d[7]) << 24);
    *reinterpret_cast<uint2*>(dst) = o;
  }
}

// One chroma sample at full resolution (x, y) from the subsampled plane p [.. x stride] with
// downsampled size dw x dh and sampling ratio fh x fv (jdsample.c; the main controller's context
// rows replicate the first / last row).
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, int stride, int dw, int dh,
                                         int fh, int fv, int x, int y) {
  if (fh == 1 && fv == 1) return p[(long)y * stride + x];
  // jdsample.c jinit_upsampler: the fancy (triangle) filters are only installed for components more than two
  // samples wide; narrower ones (images up to 4 pixels wide) are upsampled by plain replication
  // (h2v1_upsample / h2v2_upsample).  Found by tools/jpeg_fuzz.py.
  if (fh == 2 && dw <= 2) return p[(long)(fv == 2 ? y >> 1 : y) * stride + (x >> 1)];
  if (fh == 2 && fv == 1) {  // h2v1_fancy_upsample
    const uint8_t* row = p + (long)y * stride;
    const int i = x >> 1;
    const int v = row[i];
    if (x == 0 || x == 2 * dw - 1) return v;
    return (x & 1) ? (3 * v + row[i + 1] + 2) >> 2 : (3 * v + row[i - 1] + 1) >> 2;
  }
  const int iy = y >> 1;
  int ny = (y & 1) ? iy + 1 : iy - 1;  // the nearer neighbour row
  ny = ny < 0 ? 0 : (ny > dh - 1 ? dh - 1 : ny);
  const uint8_t* r0 = p + (long)iy * stride;
  const uint8_t* r1 = p + (long)ny * stride;
  if (fh == 1) {  // h1v2_fancy_upsample
    return (y & 1) ? (3 * r0[x] + r1[x] + 2) >> 2 : (3 * r0[x] + r1[x] + 1) >> 2;
  }
  // h2v2_fancy_upsample
  const int i = x >> 1;
  const int cur = 3 * r0[i] + r1[i];
  if (x == 0) return (cur * 4 + 8) >> 4;
  if (x == 2 * dw - 1) return (cur * 4 + 7) >> 4;
  if (x & 1) return (cur * 3 + (3 * r0[i + 1] + r1[i + 1]) + 7) >> 4;
  return (cur * 3 + (3 * r0[i - 1] + r1[i - 1]) + 8) >> 4;
}

__device__ __forceinline__ uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ __launch_bounds__(256) void jpeg_rgb_kernel(FrameDev f, const uint8_t* __restrict__ planes,
                                                       uint8_t* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= f.width) return;
  const int yv = planes[f.plane_off[0] + (long)y * f.bx[0] * 8 + x];
  uint8_t* o = out + ((long)y * f.width + x) * 3;
  if (f.ncomp == 1) {
    o[0] = o[1] = o[2] = (uint8_t)yv;
    return;
  }
  const int fh = f.hmax / f.h[1], fv = f.vmax / f.v[1];
  const int dw = (f.width * f.h[1] + f.hmax - 1) / f.hmax;   // compptr->downsampled_width
  const int dh = (f.height * f.v[1] + f.vmax - 1) / f.vmax;
  const int cb = chroma_at(planes + f.plane_off[1], f.bx[1] * 8, dw, dh, fh, fv, x, y) - 128;
  const int cr = chroma_at(planes + f.plane_off[2], f.bx[2] * 8, dw, dh, fh, fv, x, y) - 128;
  // jdcolor.c build_ycc_rgb_table (SCALEBITS 16) folded into the arithmetic
  o[0] = clamp255(yv + ((91881 * cr + 32768) >> 16));
  o[1] = clamp255(yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
  o[2] = clamp255(yv + ((116130 * cb + 32768) >> 16));
}

}  // namespace

hipError_t launch_jpeg_reconstruct(const JpegFrame& fr, const int16_t* d_coefs, uint8_t* d_planes,
                                   uint8_t* d_out_hwc, hipStream_t s) {
  FrameDev f{};
  f.width = fr.width; f.height = fr.height; f.ncomp = fr.ncomp; f.hmax = fr.hmax; f.vmax = fr.vmax;
  long total = 0;
  for (int c = 0; c < 3; ++c) {
    f.h[c] = fr.h[c]; f.v[c] = fr.v[c]; f.bx[c] = fr.bx[c]; f.by[c] = fr.by[c];
    f.coef_off[c] = fr.coef_off[c]; f.plane_off[c] = fr.plane_off[c];
    f.nblocks[c] = c < fr.ncomp ? (long)fr.bx[c] * fr.by[c] : 0;
    total += f.nblocks[c];
  }
  memcpy(f.q, fr.q, sizeof(f.q));
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, s, f, d_coefs,
                     d_planes, total);
  hipLaunchKernelGGL(jpeg_rgb_kernel, dim3((fr.width + 255) / 256, fr.height), dim3(256), 0, s, f,
                     d_planes, d_out_hwc);
  return hipGetLastError();
}

}  // namespace oake
