// The LOGICAL filter bank F[tap][ci][co] (tap = r*3+s, ci < Cin, co < Cout) that the 3x3 halo kernels contract against, for the
// orientations (`transpose_flip`) a layer's HWIO bank w can be used in.  Shared by the 16-bit plane images of convsplit.hip
// (ladder_filter_pack_split, prec = f16x3 / bf16x3 / bf16x6) and the fp32 banks of convf32.hip (prec = f32), so that both precisions
// contract against the SAME fp32 values.
//
//   0  forward                 F[tap][ci][co] = w[tap][ci][co]                (w = [ntaps][Cin][Cout])
//   1  backward-data           F[tap][ci][co] = w[ntaps-1-tap][co][ci]        (w = [ntaps][Cout][Cin]: flipped and transposed)
//   2  backward-data of a 3x3 / stride-2 / SAME conv on an even map as ONE stride-1 correlation over dy: the four output-parity classes
//      (h % 2, w % 2) of dx are four blocks of Cout / 4 output columns; class (ph, pw) uses tap (a, b) iff a in {1} u {0 if ph == 0},
//      b likewise, and that tap carries w[r][s] with r = (a == 1 ? ph : 2), s = (b == 1 ? pw : 2)   (w = [3][3][Cout/4][Cin])
//   3  forward of (factor-2 legacy-bilinear upsample -> 3x3 SAME conv) over the LOW-resolution map: up[2i] = x[i], up[2i+1] = (x[i]+x[i+1])/2,
//      so output-parity class (a, b) = (row % 2, col % 2) is a 3x3 correlation over x itself with
//         W_eff[a,b][dr][dc] = sum_{r,s} A_a[dr][r] A_b[dc][s] w[r][s],  A_0 = [[1/2,0,0],[1/2,1,1/2],[0,0,1/2]],  A_1 = [[0,0,0],[1,1/2,0],[0,1/2,1]]
//      (9 / 6 / 6 / 4 non-zero taps = 25 instead of 36 products per 2x2 output block); classes = four blocks of Cout / 4 output columns
//      (w = [3][3][Cin][Cout/4])
//   4  backward-data of that pair: dx_lo[p] = sum_{k=-2..2} G_k^T dy[2p+k] per axis = a 3x3 correlation over the four pixel-parity
//      classes of dy taken as four blocks of Cin / 4 INPUT channels (class a holds the taps k = 2 (dr - 1) + a; coefficient of w[r] in
//      tap dr of class a = A_a[2 - dr][r])   (w = [3][3][Cout][Cin/4])
//   5  forward of a 3x3 / stride-2 / SAME conv on an even map (TF padding: none above / left, one line below / right) as ONE stride-1
//      correlation over the four pixel-parity classes of x taken as four blocks of Cin / 4 INPUT channels: class (a, b) holds x[2i+a, 2j+b];
//      tap (dr, dc) reads class pixel (i + dr - 1, j + dc - 1) and carries w[2 (dr - 1) + a][2 (dc - 1) + b] -- rows dr = 1 (both classes)
//      and dr = 2 (class a = 0 only), columns alike: 4 / 2 / 2 / 1 taps   (w = [3][3][Cin/4][Cout], strict fp32 only)
//   6  "project, then upsample" (csrc/upproj.hip), forward operand: the nine taps side by side as ONE [Cin][9 C] matrix (ntaps = 1, Cout = 9 C):
//      F[0][ci][t C + co] = w[t][ci][co]   (w = [3][3][Cin][C], strict fp32 only)
//   7  ... and its transpose, the backward-data operand [9 C][Cin_layer] (ntaps = 1, Cin = 9 C, Cout = the layer's Cin):
//      F[0][t C + co][ci] = w[t][ci][co]   (w = [3][3][Cout][C], strict fp32 only)
// Sums have <= 9 terms with power-of-two weights, evaluated in fp32 in a fixed order.
#pragma once

__device__ __forceinline__ float filter_bank_element(const float* __restrict__ w, int ntaps, int Cin, int Cout, int transpose_flip,
                                                     int tap, int ci, int co) {
  if (transpose_flip == 0) return w[((size_t)tap * Cin + ci) * Cout + co];
  if (transpose_flip == 1) return w[((size_t)(ntaps - 1 - tap) * Cout + co) * Cin + ci];
  if (transpose_flip == 2) {
    const int C = Cout >> 2, cls = co / C, cc = co - cls * C, ph = cls >> 1, pw = cls & 1, a = tap / 3, b = tap - 3 * a;
    const bool va = a == 1 || (a == 0 && ph == 0), vb = b == 1 || (b == 0 && pw == 0);
    if (!(va && vb)) return 0.f;
    const int r = a == 1 ? ph : 2, sx = b == 1 ? pw : 2;
    return w[(((size_t)r * 3 + sx) * C + cc) * Cin + ci];
  }
  if (transpose_flip == 6) {
    const int C = Cout / 9, t = co / C;
    return w[((size_t)t * Cin + ci) * C + (co - t * C)];
  }
  if (transpose_flip == 7) {
    const int C = Cin / 9, t = ci / C;
    return w[((size_t)t * Cout + co) * C + (ci - t * C)];
  }
  if (transpose_flip == 5) {
    const int C = Cin >> 2, cls = ci / C, cc = ci - cls * C, a = cls >> 1, b = cls & 1, dr = tap / 3, dc = tap - 3 * dr;
    const bool va = dr == 1 || (dr == 2 && a == 0), vb = dc == 1 || (dc == 2 && b == 0);
    if (!(va && vb)) return 0.f;
    return w[(((size_t)(2 * (dr - 1) + a) * 3 + (2 * (dc - 1) + b)) * C + cc) * Cout + co];
  }
  const float A0[3][3] = {{0.5f, 0.f, 0.f}, {0.5f, 1.f, 0.5f}, {0.f, 0.f, 0.5f}}, A1[3][3] = {{0.f, 0.f, 0.f}, {1.f, 0.5f, 0.f}, {0.f, 0.5f, 1.f}};
  float f = 0.f;
  if (transpose_flip == 3) {
    const int C = Cout >> 2, cls = co / C, cc = co - cls * C, a = cls >> 1, b = cls & 1, dr = tap / 3, dc = tap - 3 * dr;
    for (int r = 0; r < 3; ++r) {
      const float ar = a ? A1[dr][r] : A0[dr][r];
      if (ar == 0.f) continue;
      for (int sx = 0; sx < 3; ++sx) {
        const float bs = b ? A1[dc][sx] : A0[dc][sx];
        if (bs != 0.f) f += (ar * bs) * w[(((size_t)r * 3 + sx) * Cin + ci) * C + cc];      // (ar * bs: exact powers of two)
      }
    }
    return f;
  }
  // transpose_flip == 4
  const int C = Cin >> 2, cls = ci / C, cc = ci - cls * C, a = cls >> 1, b = cls & 1, dr = tap / 3, dc = tap - 3 * dr;
  for (int r = 0; r < 3; ++r) {
    const float ar = a ? A1[2 - dr][r] : A0[2 - dr][r];
    if (ar == 0.f) continue;
    for (int sx = 0; sx < 3; ++sx) {
      const float bs = b ? A1[2 - dc][sx] : A0[2 - dc][sx];
      if (bs != 0.f) f += (ar * bs) * w[(((size_t)r * 3 + sx) * Cout + co) * C + cc];
    }
  }
  return f;
}

// Tap masks (9 bits per class, class c in bits [9c, 9c+9)) of the three class-structured banks: which taps are non-zero.
static inline unsigned long long filter_bank_tap_masks(int transpose_flip) {
  unsigned long long m = 0;
  for (int cls = 0; cls < 4; ++cls) {
    const int a = cls >> 1, b = cls & 1;
    unsigned t = 0;
    for (int dr = 0; dr < 3; ++dr)
      for (int dc = 0; dc < 3; ++dc) {
        bool on = true;
        if (transpose_flip == 2) on = (dr == 1 || (dr == 0 && a == 0)) && (dc == 1 || (dc == 0 && b == 0));
        else if (transpose_flip == 3) on = (a == 0 || dr >= 1) && (b == 0 || dc >= 1);
        else if (transpose_flip == 4) on = (a == 0 || dr <= 1) && (b == 0 || dc <= 1);
        else if (transpose_flip == 5) on = (dr == 1 || (dr == 2 && a == 0)) && (dc == 1 || (dc == 2 && b == 0));
        if (on) t |= 1u << (dr * 3 + dc);
      }
    m |= (unsigned long long)t << (9 * cls);
  }
  return m;
}
