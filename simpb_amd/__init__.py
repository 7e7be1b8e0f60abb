"""MI355X (gfx950) implementation of SimPB's hybrid 2D/3D decoder hot path behind the mmdet3d_plugin API."""
import os as _os

# Vendor convolutions beside our kernels. The gfx950 double-K 16-bit matrix instructions (v_mfma_f32_32x32x16_f16 and its
# three siblings) make OTHER kernels' vector arithmetic go wrong while they execute (DESIGN.md section 4,
# profiles/r02_mfma_x16_interference/). Our own kernels do not issue them; MIOpen's composable-kernel (CK) XDL convolution
# solvers do: with them enabled MIOpen picked kernel_grouped_conv_fwd_multiple_abd_xdl_cshuffle for the FPN's 3x3 output
# convolutions and daf_fwd_rows beside that kernel returned wrong channels in 595-599 of 600 launches; with these two
# solver families off it picks its assembly implicit-GEMM kernels (igemm_fwd_gtcx35_nhwc_fp16 ... wt32x32x8: the older
# instruction) and the same stress shows 0 of 600 (tools/daf_stress.py --co conv:256,256,32,88,3,1). The variables are
# read by MIOpen when it first looks for a convolution solver, so they are set at import; a value the user exported wins.
# Since round 3 a frame of the product path calls no vendor convolution at all (own stem, csrc/stem.hip; a GPU test patches
# F.conv2d to raise); the switches below matter for the cross-check routes of the tests and for callers that run the
# PyTorch statement of the backbone (routes.conv*_kernel = False) beside the decoder.
for _name in ("MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_FWD_XDLOPS", "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_FWD_XDLOPS"):
    if _os.environ.setdefault(_name, "0") != "0":
        import warnings as _warnings
        _warnings.warn(f"{_name}={_os.environ[_name]} (set outside simpb_amd): MIOpen may pick composable-kernel XDL convolution "
                       "solvers, whose double-K matrix instructions corrupt other kernels running beside them on gfx950 "
                       "(DESIGN.md section 4)", RuntimeWarning)
