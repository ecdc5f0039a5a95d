"""hipGraph capture of the FF-RAFT forward (inference).

One forward is ~700 small launches (8.8 ms of host time); at small batch the GPU finishes sooner than
the host can issue them.  Every libfocusflow_hip entry point only enqueues work on the current stream
(no allocation, no synchronisation), so the whole step is capturable: torch.cuda.CUDAGraph is a
hipGraph on ROCm and its graph-aware allocator provides the intermediate buffers."""
import torch

from . import ops


class GraphedForward:
    """Capture `model(image1, image2, mask1, mask2, raft_iters, test_mode=True)` for fixed shapes.

    Call it with new inputs of the same shape: they are copied into the captured input buffers, the
    graph is replayed and the (static) output tensors are returned — valid until the next call."""

    def __init__(self, model, example_inputs, raft_iters=12, warmup=3):
        if model.training:
            raise ValueError("capture the eval-mode forward (training mutates BatchNorm buffers and the tape)")
        self.model, self.iters = model, raft_iters
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):      # packs weights, sets kernel attributes, warms the allocator
                model(*self.static_in, raft_iters=raft_iters, test_mode=True)
        torch.cuda.current_stream().wait_stream(side)
        ops.guard_check(sync=True)          # the warm-up forwards decided the routes (exact context convolutions or not): capture those
        # the always-on range guard inside a graph: the probes go to a static pair of words (zeroed by a captured fill), which
        # every replay copies to the host for the asynchronous look at the next call
        self._guard_words = torch.zeros(2, dtype=torch.int32, device=self.static_in[0].device) if ops.RANGE_GUARD else None
        st = ops._guard_state()
        st["capture_words"] = self._guard_words
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph), torch.no_grad():
                self.out = model(*self.static_in, raft_iters=raft_iters, test_mode=True)
        finally:
            st["capture_words"] = None

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        ops.guard_check()                   # (raises if an earlier replay left the split formats' range)
        self.graph.replay()
        if self._guard_words is not None:
            ops.guard_queue(self._guard_words, "GraphedForward replay", getattr(self.model, "flow_net", None))
        return self.out

    def check_range(self):
        """Wait for the replays issued so far and raise if one of them left the fp16-split formats' range (ops: the always-on guard)."""
        ops.guard_check(sync=True)
