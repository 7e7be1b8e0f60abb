"""Host side of csrc/mlp_chain.hip: turns the reference's nn.Sequential stacks ([Linear, ReLU]*,
LayerNorm, ..., Linear, Scale) into the argument block of simpb_mlp_chain_forward. The module tree
(and so the state_dict) is untouched; the kernel reads the Linear weights through transposed
copies that are refreshed whenever a parameter is replaced or modified in place."""
import ctypes

import torch
import torch.nn as nn

from .. import _lib
from . import routes
from .layers import Scale
from .ops import _stream

MAX_OPS, MAX_CHAINS = 12, 8
POST_NONE, POST_REFINE3D, POST_REFINE2D, POST_SIGMOID = 0, 1, 2, 3
# routes.chain_rows4 (shipped): 4-row workgroups on the 4x4 matrix blocks, weights k4-packed [in/4][out][4]
# (csrc/mlp_chain.hip: mlp_chain_r4_kernel): 225 workgroups for 900 rows instead of 57. Off: the 16-row matrix-core kernel
# on the weights as stored, or (routes.chain_transposed) the VALU kernel on transposed copies.
# Launches with at least WIDE_ROWS rows (a batch of camera streams: runner independent_streams) take the 32-row kernel on the
# 32x32 matrix tiles (mlp_chain_r32_kernel, fragment-packed weights): at ~9 k rows the 4-row kernel sits at the issue rate of
# its 4x4 matrix instruction, a quarter of the fp32 matrix rate.
WIDE_ROWS = 4096
LINEAR, LAYERNORM = 0, 1
IN_ROWS, IN_SINE2D, IN_ROWS_LN = 0, 1, 2


class _Op(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("in_dim", ctypes.c_int), ("out_dim", ctypes.c_int), ("relu", ctypes.c_int),
                ("w", ctypes.c_void_p), ("b", ctypes.c_void_p)]


class _Chain(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("x2", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("out_scale", ctypes.c_void_p), ("ldx", ctypes.c_int), ("ldx2", ctypes.c_int), ("ldo", ctypes.c_int),
                ("in_dim", ctypes.c_int), ("in_mode", ctypes.c_int), ("n_ops", ctypes.c_int),
                ("post", ctypes.c_int), ("ldres", ctypes.c_int), ("res_cols", ctypes.c_int), ("div_rows", ctypes.c_int),
                ("div_col0", ctypes.c_int), ("reserved", ctypes.c_int), ("res", ctypes.c_void_p), ("div", ctypes.c_void_p),
                ("ops", _Op * MAX_OPS), ("ln_w", ctypes.c_void_p), ("ln_b", ctypes.c_void_p), ("ln_out", ctypes.c_void_p),
                ("ld_ln_out", ctypes.c_int), ("reserved2", ctypes.c_int)]


class _Args(ctypes.Structure):
    _fields_ = [("num_rows", ctypes.c_int), ("num_chains", ctypes.c_int), ("weights_transposed", ctypes.c_int),
                ("reserved", ctypes.c_int), ("chain", _Chain * MAX_CHAINS), ("m_live", ctypes.c_void_p)]


class ChainPlan:
    """One nn.Sequential parsed into kernel ops. Holds the transposed weights."""

    def __init__(self, seq):
        self.seq = seq
        self.ops = []  # (type, in, out, relu, w_tensor_getter, b_tensor_getter)
        self.scale = None
        self._wt = {}
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                self.ops.append((LINEAR, m.in_features, m.out_features, int(relu), m))
                i += 2 if relu else 1
            elif isinstance(m, nn.LayerNorm):
                if len(m.normalized_shape) != 1 or abs(m.eps - 1e-5) > 1e-12:
                    raise ValueError("unsupported LayerNorm")
                self.ops.append((LAYERNORM, m.normalized_shape[0], m.normalized_shape[0], 0, m))
                i += 1
            elif isinstance(m, Scale) and i == len(mods) - 1:
                self.scale = m
                i += 1
            else:
                raise ValueError(f"mlp_chain cannot express {type(m).__name__}")
        if len(self.ops) > MAX_OPS or any(o[1] > 256 or o[2] > 256 for o in self.ops):
            raise ValueError("chain too long or too wide for mlp_chain")
        self.in_dim = self.ops[0][1]
        self.out_dim = self.ops[-1][2]

    def _transposed(self, lin):
        key = id(lin)
        w = lin.weight
        tag = (w.data_ptr(), w._version, w.device)
        hit = self._wt.get(key)
        if hit is None or hit[0] != tag:
            hit = (tag, w.detach().t().contiguous().float())
            self._wt[key] = hit
        return hit[1]

    def _packed(self, lin):
        """Wp[k // 4][n][k % 4] = W[n][k] (f32, contiguous), refreshed when the parameter is replaced or modified."""
        key = ("k4", id(lin))
        w = lin.weight
        tag = (w.data_ptr(), w._version, w.device)
        hit = self._wt.get(key)
        if hit is None or hit[0] != tag:
            n, k = w.shape
            hit = (tag, w.detach().float().t().reshape(k // 4, 4, n).permute(0, 2, 1).contiguous())
            self._wt[key] = hit
        return hit[1]

    def _fragments(self, lin):
        """Wq[t][c][q][32 h + r][e] = W[32 t + r][32 c + 16 h + 4 q + e] (f32, zeros past out_dim): the B operand of
        v_mfma_f32_32x32x2f32 for tile t and 32-deep chunk c as four coalesced 1-KiB wave loads (csrc/mlp_chain.hip)."""
        key = ("frag", id(lin))
        w = lin.weight
        tag = (w.data_ptr(), w._version, w.device)
        hit = self._wt.get(key)
        if hit is None or hit[0] != tag:
            n, k = w.shape
            tiles = -(-n // 32)
            full = torch.zeros(tiles * 32, k, device=w.device, dtype=torch.float32)
            full[:n] = w.detach().float()
            # [t, r, c, h, q, e] -> [t, c, q, h, r, e]
            hit = (tag, full.reshape(tiles, 32, k // 32, 2, 4, 4).permute(0, 2, 4, 3, 1, 5).contiguous())
            self._wt[key] = hit
        return hit[1]

    def fill(self, chain, keep, wide=False):
        chain.n_ops = len(self.ops)
        for j, (typ, din, dout, relu, m) in enumerate(self.ops):
            op = chain.ops[j]
            op.type, op.in_dim, op.out_dim, op.relu = typ, din, dout, relu
            if typ == LINEAR:
                if wide and din % 32 == 0:
                    wq = self._fragments(m)
                    keep.append(wq)
                    op.w = wq.data_ptr()
                elif wide:
                    w = m.weight
                    if not w.is_contiguous() or w.dtype != torch.float32:
                        raise ValueError("mlp_chain reads nn.Linear weights in place: contiguous f32 expected")
                    op.w = w.data_ptr()
                elif routes.R.chain_rows4 and din % 4 == 0:
                    wp = self._packed(m)
                    keep.append(wp)
                    op.w = wp.data_ptr()
                elif routes.R.chain_transposed and not routes.R.chain_rows4:
                    wt = self._transposed(m)
                    keep.append(wt)
                    op.w = wt.data_ptr()
                else:
                    w = m.weight
                    if not w.is_contiguous() or w.dtype != torch.float32:
                        raise ValueError("mlp_chain reads nn.Linear weights in place: contiguous f32 expected")
                    op.w = w.data_ptr()
                op.b = m.bias.data_ptr() if m.bias is not None else None
            else:
                op.w, op.b = m.weight.data_ptr(), m.bias.data_ptr()
        chain.out_scale = self.scale.scale.data_ptr() if self.scale is not None else None


def plan_of(seq):
    plan = getattr(seq, "_simpb_plan", None)
    if plan is None:
        plan = ChainPlan(seq)
        object.__setattr__(seq, "_simpb_plan", plan)
    return plan


def _rows(t, width):
    """(pointer-ready tensor, row stride) for a [..., width] tensor whose rows are `width`
    contiguous floats at a constant stride."""
    if t.dtype != torch.float32:
        t = t.float()
    if t.stride(-1) != 1:
        t = t.contiguous()
    flat = t.reshape(-1, t.shape[-1]) if t.is_contiguous() else None
    if flat is None:
        t = t.contiguous()
        flat = t.reshape(-1, t.shape[-1])
    return flat, flat.stride(0)


def run_chains(jobs, num_rows, device, m_live=None):
    """jobs: list of dicts(plan, x=(tensor2d, ld, col), x2=(tensor2d, ld, col) or None, out=(tensor2d, ld, col),
    sine=bool, post=dict(kind, res=(tensor2d, ld), res_cols, div=tensor or None, div_rows, div_col0) or None,
    ln=(nn.LayerNorm over x, (out tensor2d, ld) or None) or None: input = LayerNorm(x) + x2, LayerNorm(x) written to out).
    All tensors f32 on `device`, 2-D views with unit inner stride. m_live: device i32 [1] or None; rows past it are capacity
    slots of the static 2D query set and come out as zeros (their workgroups do no work)."""
    if not jobs or len(jobs) > MAX_CHAINS:
        raise ValueError("1..8 chains per launch")
    args = _Args()
    args.num_rows, args.num_chains = int(num_rows), len(jobs)
    args.m_live = m_live.data_ptr() if m_live is not None else None
    # (not the sine chains: their prologue is transcendental work per element, which 32-row workgroups serialise in front
    # of the matrix work -- 2D encoder at 8.9 k rows: 84 us against 64 us on the 4-row kernel)
    wide = routes.R.chain_rows4 and int(num_rows) >= WIDE_ROWS and not any(j.get("sine") for j in jobs)
    args.weights_transposed = 3 if wide else 2 if routes.R.chain_rows4 else (1 if routes.R.chain_transposed else 0)
    keep = []
    for c, job in enumerate(jobs):
        ch = args.chain[c]
        plan = job["plan"]
        xt, ldx, xcol = job["x"]
        ch.x = xt.data_ptr() + 4 * xcol
        ch.ldx = ldx
        if job.get("x2") is not None:
            x2t, ldx2, x2col = job["x2"]
            ch.x2, ch.ldx2 = x2t.data_ptr() + 4 * x2col, ldx2
        else:
            ch.x2, ch.ldx2 = None, 0
        ot, ldo, ocol = job["out"]
        ch.out, ch.ldo = ot.data_ptr() + 4 * ocol, ldo
        ch.in_mode = IN_SINE2D if job.get("sine") else IN_ROWS
        ch.in_dim = plan.in_dim
        if job.get("ln") is not None:
            ln, ln_out = job["ln"]
            if not routes.R.chain_rows4 or ln.normalized_shape != (plan.in_dim,) or abs(ln.eps - 1e-5) > 1e-12 or job.get("sine"):
                raise ValueError("a leading LayerNorm needs the 4-row chain kernel and a norm of the chain's input width")
            ch.in_mode = IN_ROWS_LN
            ch.ln_w, ch.ln_b = ln.weight.data_ptr(), ln.bias.data_ptr()
            if ln_out is not None:
                ch.ln_out, ch.ld_ln_out = ln_out[0].data_ptr(), ln_out[1]
                keep.append(ln_out[0])
        post = job.get("post")
        if post:
            ch.post = post["kind"]
            if post.get("res") is not None:
                rt, ldres = post["res"]
                ch.res, ch.ldres, ch.res_cols = rt.data_ptr(), ldres, post["res_cols"]
                keep.append(rt)
            if post.get("div") is not None:
                ch.div, ch.div_rows, ch.div_col0 = post["div"].data_ptr(), post["div_rows"], post["div_col0"]
                keep.append(post["div"])
        plan.fill(ch, keep, wide)
        keep += [xt, ot]
    status = _lib.lib().simpb_mlp_chain_forward(ctypes.byref(args), _stream())
    _lib.check(status, "simpb_mlp_chain_forward")


def chain_forward(seq, x, x2=None, sine=False, post=None, m_live=None):
    """y = post(seq(x + x2)) for x [..., in_dim] on the GPU, one launch."""
    plan = plan_of(seq)
    xf, ldx = _rows(x, x.shape[-1])
    job = dict(plan=plan, x=(xf, ldx, 0), sine=sine, post=post)
    if x2 is not None:
        x2f, ldx2 = _rows(x2, x2.shape[-1])
        job["x2"] = (x2f, ldx2, 0)
    n = xf.shape[0]
    out = torch.empty(n, plan.out_dim, device=x.device, dtype=torch.float32)
    job["out"] = (out, plan.out_dim, 0)
    if n:
        run_chains([job], n, x.device, m_live=m_live)
    return out.reshape(x.shape[:-1] + (plan.out_dim,))
