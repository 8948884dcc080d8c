// gfx950 (MI355X): entry point of the 3x3 stride-1 convolutions in Winograd form F(4x4, 3x3) on the fp32 matrix pipe.
//
// Replaces the same torch conv2d dispatches as kernels_conv_wino.hip (reference call sites baseline/predict.py:109 -> :45-52: the
// stride-1 subunits of the ResidualUnits).  Y = A^T [ (G g G^T) . (B^T d B) ] A with a 6x6 input patch per 4x4 output tile: 36
// multiplies per 16 outputs = 2.25 per output pixel and (cin, cout) pair, where F(2x2, 3x3) needs 4 and a direct convolution 9.  All
// arithmetic is fp32 (fp32 MFMAs + fp32 VALU transforms, weights transformed in fp64 and rounded once at pack time).  The larger
// transform constants cost precision: logits land 2.1e-5 ... 3.8e-5 from an fp64 evaluation (direct fp32: 5e-6 ... 8e-6;
// tests/diagnostics/winograd_f4_precision.py) - inside the 1e-4 budget of the parity tests.
//
// The kernels live in kernels_conv_wino4x.hip (three forms, wino4_form()).  This file keeps what decides whether a layer takes them:
// an inference packing exists (engine.hip: ww4_fresh), >= 32 input channels, channel counts in multiples of 32, and a map of at least
// ONE full 16 x 16 sub-tile (round 5: on smaller maps most of a block's 16 Winograd tiles are padding, and behind an InstanceNorm over
// a handful of pixels the form's rounding is amplified - a 32 x 32 window on the 5-level net, 2 x 2 pixels at the bottom, came out
// 1.3e-4 from the oracle in the randomised sweep; such layers run on the F(2x2) kernel).  The round-3 kernel on 16x16x4 MFMAs that
// used to live here was superseded for every such layer and was removed in round 5 (docs/DESIGN_history_r1-r4.md, 3.1a').
#include "conv_common.h"
#include "pack_jobs.h"

namespace cs {

size_t packed_wino4_floats(int cout_total, int cin_total) { return (size_t)(cin_total / 16) * 36 * (cout_total / 16) * 256; }

bool wino4_fits(int cin, int cout, int h, int w) { return cin >= WINO4_MIN_CIN && cin % 32 == 0 && cout % 32 == 0 && h >= 16 && w >= 16; }

int launch_conv_wino4(hipStream_t s, const ConvSrc& a, const float* wpack_wino4, const float* bias, float* out, int pitch_out, int cout,
                      double* stats, int n, int h, int w, const ConvOutExtra* ex) {
    CS_REQUIRE(wino4_form(a.c, cout) != 0, "conv(winograd 4x4): channels (%d -> %d) have no F(4x4) kernel (wino4_fits)", a.c, cout);
    return launch_conv_wino4x(s, a, wpack_wino4, bias, out, pitch_out, cout, stats, n, h, w, ex);
}

}  // namespace cs
