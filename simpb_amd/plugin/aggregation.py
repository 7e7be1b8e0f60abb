"""models/aggregation.py of the reference: AdaptiveQueryAggregation + ReWeight."""
import torch
import torch.nn as nn

from . import dense, routes
from .allocation import Allocation2D, aggregate_2d_to_3d
from .registry import ATTENTION, PLUGIN_LAYERS, build_from_cfg


class ReWeight(nn.Module):
    """aggregation.py:10-40 (trans=True branch): only the alpha MLP lives here; the weighted mean
    itself is the simpb_aggregate_2d_to_3d kernel."""

    def __init__(self, c_dim, f_dim=256, trans=True, with_pos=False):
        super().__init__()
        self.c_dim, self.f_dim, self.trans, self.with_pos = c_dim, f_dim, trans, with_pos
        self.reduce = nn.Sequential(nn.Linear(c_dim, f_dim), nn.ReLU())
        self.alpha = nn.Sequential(nn.Linear(f_dim, 1), nn.Sigmoid())

    def forward(self, parameter):
        return self.alpha(self.reduce(parameter))


@PLUGIN_LAYERS.register_module()
class AdaptiveQueryAggregation(nn.Module):
    def __init__(self, self_attn=None, reweight=None, decouple_attn=False, with_pos=False):
        super().__init__()
        if self_attn is None or reweight is None or not with_pos:
            raise NotImplementedError("the SimPB configs use self_attn + reweight + with_pos")
        self.with_pos = with_pos
        self.decouple_attn = decouple_attn
        self.reweight = ReWeight(c_dim=257, trans=True, with_pos=with_pos)
        self.self_attn = build_from_cfg(self_attn, ATTENTION)

    def forward(self, query2d, query_pos2d, anchor2d, query3d, query_pos3d, anchor3d, dn_query2d=None,
                dn_query_pos2d=None, dn_anchor2d=None, dn_query3d=None, dn_query_pos3d=None, dn_anchor3d=None,
                trans_matrix=None, center_matrix=None, dn_trans_matrix=None, dn_center_matrix=None, attn_mask=None,
                graph_model=None, allocation=None, **kwargs):
        """aggregation.py:54-101, eval path. `allocation` (an Allocation2D) carries the index form;
        without it the dense matrices are converted (one-hot rows -> indices)."""
        if dn_query2d is not None or dn_query3d is not None or attn_mask is not None:
            raise NotImplementedError("denoising queries only exist in training")
        if allocation is None:
            allocation = _from_dense(trans_matrix, center_matrix)
        a2q, shape3d = allocation.a2q, query3d.shape
        if getattr(allocation, "streams", 0):
            # a batch of independent streams: the 2D set is one flat slot array and a2q holds flat slots, so the 3D side
            # joins it as [1, bs * N3, .] views (allocation.allocate_independent)
            a2q = a2q.reshape(1, -1, a2q.shape[-1])
            query3d, query_pos3d = query3d.reshape(1, -1, shape3d[-1]), query_pos3d.reshape(1, -1, shape3d[-1])
        if routes.R.dense and query2d.is_cuda and query2d.shape[-1] % 64 == 0 and allocation.is_center.dtype == torch.int32:
            # ReWeight (:10-40) in two launches: reduce over cat(query2d, is_center) as one GEMM whose
            # 257th input column is a flagged extra bias, then the alpha row-dot + sigmoid
            m_live = kwargs.get("m_live")
            red, alp = self.reweight.reduce[0], self.reweight.alpha[0]
            w_x, w_flag = dense.fold_split_last_column(red)
            hidden = dense.linear(query2d, w_x, red.bias, relu=True, m_live=m_live,
                                  row_flag=allocation.is_center.contiguous().reshape(-1), bias2=w_flag)
            if a2q.shape[-1] <= 8:   # alpha = sigmoid(alp(hidden)) inside the aggregation launch
                query3d, query_pos3d = aggregate_2d_to_3d(query3d, query_pos3d, query2d, query_pos2d, None, a2q,
                                                          hidden=hidden, alpha_fc=alp)
                query3d, query_pos3d = query3d.reshape(shape3d), query_pos3d.reshape(shape3d)
                aggregated = graph_model(self.self_attn, query=query3d, query_pos=query_pos3d, attn_mask=attn_mask)
                return aggregated, query_pos3d, anchor3d
            alpha = dense.rowdot_sigmoid(hidden, alp.weight, alp.bias, m_live=m_live)
        else:
            center_param = torch.cat([query2d, allocation.is_center[..., None].to(query2d.dtype)], dim=-1)
            alpha = self.reweight(center_param)
        query3d, query_pos3d = aggregate_2d_to_3d(query3d, query_pos3d, query2d, query_pos2d, alpha, a2q)
        query3d, query_pos3d = query3d.reshape(shape3d), query_pos3d.reshape(shape3d)
        aggregated = graph_model(self.self_attn, query=query3d, query_pos=query_pos3d, attn_mask=attn_mask)
        return aggregated, query_pos3d, anchor3d


def _from_dense(trans_matrix, center_matrix):
    """Dense one-hot [bs, N2, N3] pair -> Allocation2D (for callers holding the reference's tensors)."""
    bs, n2, n3 = trans_matrix.shape
    out = Allocation2D()
    has = trans_matrix.sum(-1) > 0
    out.q2a = torch.where(has, trans_matrix.argmax(-1), -1).to(torch.int32).contiguous()
    out.is_center = center_matrix.sum(-1).to(torch.int32).contiguous()
    out.num_anchor = n3
    # (anchor, k) -> its k-th slot in ascending slot order; an anchor has at most one slot per camera
    t = trans_matrix.permute(0, 2, 1) > 0
    rank = t.cumsum(-1) - 1
    width = max(int(t.sum(-1).max()), 1)
    a2q = torch.full((bs, n3, width), -1, dtype=torch.int32, device=trans_matrix.device)
    b, a, slot = torch.nonzero(t, as_tuple=True)
    a2q[b, a, rank[b, a, slot]] = slot.to(torch.int32)
    out.a2q = a2q
    out.query_cam = None
    out.query_groups = None
    out.count = None
    return out
