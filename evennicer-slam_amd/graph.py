"""HIP-graph capture of a whole optimisation step (render + caller's loss + backward).

The reference's loops issue several hundred tiny device launches per iteration from Python; on MI355X the
kernels of this path finish faster than Python can enqueue them.  Capturing the step once (torch.cuda.graphs
drives hipGraph on ROCm; the library's launches are ordinary stream-ordered launches and are captured with
everything else) makes an iteration a single hipGraphLaunch.

Requirements on `step_fn`: it reads and writes only tensors that stay allocated (static inputs: update them
in place with .copy_ between replays), performs no host synchronisation (.item(), boolean-mask indexing), and
leaves gradients in `.grad` of the leaves.  The ray count is fixed at capture time.  No tensor of an EARLIER
eager step that still carries an autograd graph (a kept loss, a kept render output) may be alive at capture time:
its backward nodes were created on another stream and the capture of the new step's backward can crash the process."""
import gc

import torch


class GraphedStep:
    def __init__(self, step_fn, warmup=3):
        self.step_fn = step_fn
        from . import functional as EF
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        EF.plans_suspended[0] += 1              # the warm-up goes the route the capture will go: its caches are the ones the graph reads
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):         # first-use work (module loading, attribute calls) stays outside
                    out = step_fn()
                    del out
        finally:
            EF.plans_suspended[0] -= 1
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gc.collect()                            # reference cycles holding tensors of the warm-up steps (see below)
        self.graph = torch.cuda.CUDAGraph()
        EF.begin_capture()
        ok = False
        try:
            with torch.cuda.graph(self.graph):
                self.out = step_fn()
            ok = True
        finally:
            # tensors the captured kernels overwrite behind torch's back (FusedAdam, MaskedGridOptimizer): the Python
            # calls that announce those writes only ran now, so every replay announces them again
            self.raw_writes = EF.end_capture(ok)
            self._persist_keys = list(getattr(EF.end_capture, 'persist_keys', []))

    def close(self):
        """Release the graph and withdraw its persistent gradient buffers from the library's registry.  Call it when the step
        is no longer replayed (the destructor does the same, but a GraphedStep caught in a reference cycle is collected late)."""
        keys, self._persist_keys = getattr(self, '_persist_keys', []), []
        if keys:
            from . import functional as EF
            EF.forget_persistent(keys)
        self.graph = None
        self.out = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def replay(self):
        if self.graph is None:
            raise RuntimeError("this GraphedStep has been closed")
        self.graph.replay()
        if self.raw_writes:
            torch._C._increment_version(self.raw_writes)        # one call for the list (a bare tensor would be iterated row by row)
        return self.out
