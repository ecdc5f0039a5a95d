"""hipGraph capture of the FF-RAFT forward (inference).

One forward is ~700 small launches (8.8 ms of host time); at small batch the GPU finishes sooner than
the host can issue them.  Every libfocusflow_hip entry point only enqueues work on the current stream
(no allocation, no synchronisation), so the whole step is capturable: torch.cuda.CUDAGraph is a
hipGraph on ROCm and its graph-aware allocator provides the intermediate buffers."""
import torch


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
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = model(*self.static_in, raft_iters=raft_iters, test_mode=True)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.out
