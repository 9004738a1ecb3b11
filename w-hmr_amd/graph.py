"""HIP-graph replay of a forward pass (the PyMAF loop is ~330 small dependent launches: at small batch the host
launch cost dominates; a captured graph replays them with one submission).

    fast = GraphedForward(model, x, None, center, scale, bbox_height, orig_shape, bbox_info, full_x=full)
    out = fast(x2, None, center2, ...)        # same shapes; inputs are copied into the captured buffers

Every kernel of libwhmr_hip.so only enqueues on the stream it is given and allocates nothing, so capture needs no
special casing; tensor allocations made by the module during capture live in the graph's private pool.
"""
import torch


class GraphedForward:
    def __init__(self, module, *args, warmup=3, **kwargs):
        self.module = module
        self._args = [a.clone() if torch.is_tensor(a) else a for a in args]
        self._kwargs = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in kwargs.items()}
        from . import _lib as L
        side = L.side_stream(torch.cuda.current_device(), 2)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up: builds weight caches, workspaces, kernel attributes
            for _ in range(warmup):
                module(*self._args, **self._kwargs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = module(*self._args, **self._kwargs)

    def __call__(self, *args, **kwargs):
        for dst, src in zip(self._args, args):
            if torch.is_tensor(dst):
                dst.copy_(src)
        for k, src in kwargs.items():
            if torch.is_tensor(self._kwargs.get(k)):
                self._kwargs[k].copy_(src)
        self.graph.replay()
        return self.out
