"""Device-resident fusion on torch-allocated HBM buffers (torch is plumbing only: memory, streams)."""
import numpy as np
import torch

from . import native


def _stream_handle(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)


class DeviceFusion:
    """T ticks x N sensors -> T merged clouds, inputs and outputs resident in HBM.

    depth: torch.int16 / uint16-bit-pattern tensor [T, sum(w*h)] (or any shape with that many u16 per tick),
    rgb:   torch.uint8 tensor [T, sum(w*h)*3].  Outputs: vertices torch.uint8 [T, capacity, 16] viewed as
    VertexC4ubV3f, offsets torch.int32 [T, N+1] (offsets[k, i] = first vertex of sensor i, offsets[k, N] = nVertices).
    """

    def __init__(self, n_ticks, widths, heights, device=None, mode=0):
        if not torch.cuda.is_available():
            raise native.NativeUtilsError("DeviceFusion needs a HIP device (no CPU path)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.plan = native.FusionPlan(self.device.index, n_ticks, widths, heights)
        self.plan.set_mode(mode)
        self.n_ticks, self.n_maps = n_ticks, self.plan.n_maps
        self.capacity = self.plan.capacity
        self.vertices = torch.empty((n_ticks, self.capacity, 16), dtype=torch.uint8, device=self.device)
        self.offsets = torch.zeros((n_ticks, self.n_maps + 1), dtype=torch.int32, device=self.device)

    @property
    def tiles_per_tick(self):
        return self.plan.tiles_per_tick

    def set_params(self, intr, wt, bounds):
        self.plan.set_params(intr, wt, bounds, _stream_handle())

    def run(self, depth, rgb, stream=None):
        assert depth.is_cuda and rgb.is_cuda and depth.is_contiguous() and rgb.is_contiguous()
        assert depth.element_size() == 2 and depth.numel() == self.n_ticks * self.plan.pixels_per_tick, depth.shape
        assert rgb.dtype == torch.uint8 and rgb.numel() == self.n_ticks * self.plan.pixels_per_tick * 3, rgb.shape
        self.plan.run(depth.data_ptr(), rgb.data_ptr(), self.vertices.data_ptr(), self.offsets.data_ptr(), _stream_handle(stream))
        return self.vertices, self.offsets

    def tick_cloud(self, k):
        """Host copy of tick k's merged cloud as a VERTEX_DTYPE array (synchronises)."""
        off = self.offsets[k].cpu().numpy()
        n = int(off[-1])
        raw = self.vertices[k, :n].cpu().numpy()
        return raw.view(native.VERTEX_DTYPE).reshape(-1), off


def upload_rig(rig, n_ticks=1, device=None):
    """Rig (synth.Rig) -> (depth int16 [T, P], rgb uint8 [T, 3P]) on the GPU, the same tick replicated T times."""
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    d = torch.from_numpy(rig.depth_maps.view(np.int16).copy()).to(dev)
    c = torch.from_numpy(rig.depth_colors.copy()).to(dev)
    return d.unsqueeze(0).repeat(n_ticks, 1).contiguous(), c.unsqueeze(0).repeat(n_ticks, 1).contiguous()
