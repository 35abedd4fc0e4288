"""Keep several forwards in flight (no reference counterpart: the reference's loop is `for batch: model(batch)`).

One launch of the fused block per batch leaves about 11 of its 88 us to things a NEIGHBOURING launch can hide: the gap
between dependent launches of one stream, the two serial memory latencies before the first group of a wave has its
rows, and the tail.  `InFlight` runs consecutive calls on alternating streams, so the device always has the next
batch's workgroups to start while the previous batch drains (fused block: 740 -> 829 M samples/s at B = 65 536,
DESIGN.md section 5).  Inputs must stay untouched until the result was waited for; results are ordinary tensors.
"""
import torch


class InFlight:
    """fl = InFlight(model, n=2);  h = fl.submit(ids, vals) ... y = fl.result(h)

    `call` selects what is run: "forward" (logits, the default) or "arm_block" (the fused block's output)."""

    def __init__(self, model, n=2, call="forward"):
        if n < 1:
            raise ValueError("n >= 1")
        self.model, self.call = model, call
        self._streams = None
        self._n, self._k = int(n), 0

    def _stream(self, device):
        if self._streams is None or self._streams[0].device != device:
            self._streams = [torch.cuda.Stream(device=device) for _ in range(self._n)]
        s = self._streams[self._k % self._n]
        self._k += 1
        return s

    def _warm(self):
        """refresh the lazily built parameter caches (q_fold + BatchNorm affines, the sibling models' folds, the re-cut
        shard, the MLPs' packed / folded weights) on the CALLER's stream: they are produced by whichever stream first
        needs them and then read by every later submit on other streams, so they must exist before the side streams fork
        off (advisor findings r2, r3: one model hook, `warm_caches`, covers ARM-Net, GC-ARM and AFN)"""
        self.model.warm_caches(heads=self.call != "arm_block")

    def submit(self, ids, vals):
        """Enqueue one batch; returns a handle for result().  x['value'] semantics: vals is clamped in place.
        The handle keeps ids / vals alive, and both are recorded on the side stream, so a caller that passes temporaries
        (`fl.submit(i, v.clone())`) cannot have their memory handed out again while the side stream still reads them."""
        self._warm()
        s = self._stream(vals.device)
        s.wait_stream(torch.cuda.current_stream(vals.device))        # the inputs were produced on the caller's stream
        ids.record_stream(s)
        vals.record_stream(s)
        with torch.no_grad(), torch.cuda.stream(s):
            y = self.model.arm_block(ids, vals) if self.call == "arm_block" else self.model({"id": ids, "value": vals})
            done = s.record_event()
        return y, done, s, (ids, vals)

    @staticmethod
    def result(handle):
        """Make the caller's current stream wait for that batch and hand out its tensor."""
        y, done = handle[0], handle[1]
        cur = torch.cuda.current_stream(y.device)
        cur.wait_event(done)
        y.record_stream(cur)
        return y
