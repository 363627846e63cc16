"""Oracle: the gcndec denoiser (models/direction_diffusion.py:17-44) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The arithmetic of torch_geometric.nn.GCNConv lives in a third-party package
that is neither vendored in the reference nor installed here (version unpinned: the reference
has no requirements file), and the reference holds no fixture for it.  This restates the
published definition (Kipf & Welling 2017; PyG docs of GCNConv / gcn_norm): remaining self
loops added with weight 1, unit edge weights, deg = in-degree over targets,
out = D^-1/2 (A + I) D^-1/2 (x W^T) + b, messages flow source -> target.
"""
import torch
import torch.nn.functional as F


def gcn_norm(edge_index, n):
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    loops = torch.arange(n)
    src, dst = torch.cat([src[keep], loops]), torch.cat([dst[keep], loops])
    deg = torch.zeros(n, dtype=torch.float64).index_add_(0, dst, torch.ones(dst.shape[0], dtype=torch.float64))
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0
    return src, dst, dis[src] * dis[dst]


def gcn_conv(x, edge_index, w, b):
    src, dst, wgt = gcn_norm(edge_index, x.shape[0])
    xw = F.linear(x, w)
    out = torch.zeros(x.shape[0], w.shape[0], dtype=x.dtype)
    out.index_add_(0, dst, wgt.to(x.dtype).unsqueeze(1) * xw[src])
    return out + b


def gcndec(x, edge_index, t, convs, fcs):
    """convs = [(W, b)] x 4 (GCNConv lin.weight / bias), fcs = [(W, b)] x 3"""
    h = torch.cat((x.to(torch.float32), t.unsqueeze(1)), dim=1).float().to(convs[0][0].dtype)
    for w, b in convs:
        h = F.gelu(gcn_conv(h, edge_index, w, b))
    for i, (w, b) in enumerate(fcs):
        h = F.linear(h, w, b)
        if i + 1 < len(fcs):
            h = F.gelu(h)
    return h
