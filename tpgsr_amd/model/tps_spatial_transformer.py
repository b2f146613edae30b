"""TPS spatial transformer constants (reference: model/tps_spatial_transformer.py:54-95).

Only the registered buffers live here (`inverse_kernel`, `padding_matrix`, `target_coordinate_repr`,
`target_control_points`: part of every TSRN checkpoint); the grid generation and bilinear sampling run as the
tpgsr_tps_grid_* / tpgsr_grid_sample_* kernels inside the TSRN plan."""
import numpy as np
import torch
from torch import nn


def build_output_control_points(num_control_points, margins):
    """10 points along the top margin, 10 along the bottom (reference :38-50)."""
    mx, my = margins
    k = num_control_points // 2
    xs = np.linspace(mx, 1.0 - mx, k)
    pts = np.concatenate([np.stack([xs, np.full(k, my)], 1), np.stack([xs, np.full(k, 1.0 - my)], 1)], 0)
    return torch.tensor(pts, dtype=torch.float32)


def compute_partial_repr(input_points, control_points):
    """U(r) = 0.5 r^2 log r^2 with 0 log 0 := 0 (reference :22-34)."""
    d = input_points[:, None, :] - control_points[None, :, :]
    d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
    r = 0.5 * d2 * torch.log(d2)
    return torch.where(torch.isnan(r), torch.zeros_like(r), r)


class TPSSpatialTransformer(nn.Module):
    def __init__(self, output_image_size=None, num_control_points=None, margins=None):
        super().__init__()
        self.output_image_size = tuple(output_image_size)
        self.num_control_points = num_control_points
        self.margins = margins
        self.target_height, self.target_width = self.output_image_size
        n = num_control_points
        tcp = build_output_control_points(n, margins)
        fk = torch.zeros(n + 3, n + 3)
        fk[:n, :n] = compute_partial_repr(tcp, tcp)
        fk[:n, n] = 1
        fk[n, :n] = 1
        fk[:n, n + 1:] = tcp
        fk[n + 1:, :n] = tcp.t()
        inverse_kernel = torch.inverse(fk)
        h, w = self.output_image_size
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        coord = torch.stack([xs.reshape(-1) / (w - 1), ys.reshape(-1) / (h - 1)], 1)
        rep = torch.cat([compute_partial_repr(coord, tcp), torch.ones(h * w, 1), coord], 1)
        self.register_buffer("inverse_kernel", inverse_kernel.contiguous())  # torch.inverse returns column-major strides
        self.register_buffer("padding_matrix", torch.zeros(3, 2))
        self.register_buffer("target_coordinate_repr", rep.contiguous())
        self.register_buffer("target_control_points", tcp.contiguous())

    def forward(self, input, source_control_points, align_corners=False):
        """(N, C, H, W), (N, num_control_points, 2) -> (rectified (N, C, th, tw), source coordinates (N, th*tw, 2))
        (reference :97-112; F.grid_sample's align_corners default of torch >= 1.3).  Inside a TSRN this is part of the fused
        plan; standalone it is the same two kernels (TPS grid with fp64 accumulation, bilinear sampler) with their adjoints."""
        from .. import functional as Fh
        if source_control_points.dim() != 3 or source_control_points.shape[1] != self.num_control_points or \
                source_control_points.shape[2] != 2:
            raise ValueError(f"expected control points (N, {self.num_control_points}, 2), got {tuple(source_control_points.shape)}")
        th, tw = self.output_image_size
        grid, src = Fh.tps_grid(source_control_points, self.inverse_kernel, self.target_coordinate_repr, th * tw)
        out = Fh.grid_sample(Fh.to_nhwc(input), grid, (th, tw), align_corners)
        return Fh.to_nchw(out), src
