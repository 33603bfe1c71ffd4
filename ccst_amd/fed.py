"""Drop-in for the federated hot path of federated/fed_run.py on HIP kernels + RCCL:

    train(model, train_loader, optimizer, loss_fun, client_num, device, args, iter_idx, logger)   fed_run.py:31-88
    test(model, test_loader, loss_fun, device, args)                                               fed_run.py:214-259
    communication(args, server_model, models, client_weights)                                       fed_run.py:385-455

plus the pieces the reference takes from torch: ``CrossEntropyLoss`` (fed_run.py:554) and ``SGD``
(fed_run.py:657), here as single HIP launches over a flat parameter arena (``FlatParams``), and
``communication_distributed`` -- the MI355X form of FedAvg: one process per GPU/client, the weighted
average is ONE all-reduce(SUM) over xGMI of the flat fp32 state.
"""
import torch
from torch import nn

from . import _lib, nn_ops, ops
from ._lib import check, ptr, stream_ptr


# ---------------------------------------------------------------------------
# flat parameter arena
# ---------------------------------------------------------------------------
class FlatParams(object):
    """Re-homes a model's float parameters and float buffers (BN running stats) into one contiguous
    fp32 arena [params | buffers], and the gradients into a second one.  SGD is then one launch, and
    FedAvg one all-reduce with no packing copies.  ``num_batches_tracked`` (int64) stays outside."""

    def __init__(self, model):
        params = [p for p in model.parameters()]
        bufs = [b for b in model.buffers() if b.dtype == torch.float32]
        if not params:
            raise ValueError("model has no parameters")
        dev = params[0].device      # arena = memory re-homing only; the kernels that use it need the GPU

        def al(n):
            return (n + 3) // 4 * 4          # keep every tensor 16-byte aligned inside the arena

        self.n_param = sum(al(p.numel()) for p in params)
        self.n_total = self.n_param + sum(al(b.numel()) for b in bufs)
        self.flat = torch.zeros(self.n_total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.n_param, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                off += al(n)
            assert off == self.n_param
            for b in bufs:
                n = b.numel()
                self.flat[off:off + n].copy_(b.reshape(-1))
                b.data = self.flat[off:off + n].view(b.shape)
                off += al(n)
        self.params = params
        self.bufs = bufs
        self.model = model
        model.__dict__["_ccst_arena"] = self
        model.__dict__.pop("_ccst_graph_steps", None)     # captured train steps hold the old tensors' addresses

    def __deepcopy__(self, memo):
        """copy.deepcopy(model) clones every parameter into storage of its own (fed_run.py:577 makes the clients that way), so a
        copied arena would alias nothing: the copy gets none and builds its own on first use."""
        return None

    def valid(self, full=False):
        """Do the model's tensors still live in this arena?  model.to('cpu') / .to(device) (fed_run.py:32,85 -- the
        ``offload_models`` path), load-by-assignment or a deepcopy replace p.data; kernels fed from a stale arena would
        silently stop training the model."""
        p0 = self.params[0]
        if p0.device != self.flat.device or p0.data_ptr() != self.flat.data_ptr():
            return False
        if full:
            base, off = self.flat.data_ptr(), 0
            for t in list(self.params) + list(self.bufs):
                if t.data_ptr() != base + 4 * off or t.device != self.flat.device:
                    return False
                off += (t.numel() + 3) // 4 * 4
            live = [p for p in self.model.parameters()]
            if len(live) != len(self.params) or any(a is not b for a, b in zip(live, self.params)):
                return False
        return True

    @staticmethod
    def of(model):
        """The model's arena, rebuilt from the tensors the modules currently hold whenever they no longer alias it."""
        a = model.__dict__.get("_ccst_arena")
        if a is None or a.model is not model or not a.valid(full=True):
            a = FlatParams(model)
        return a

    def require_cuda(self, what):
        if not self.flat.is_cuda:
            raise RuntimeError("ccst_amd.fed: %s runs on the GPU and the model is on %s; move it with model.to(device) first "
                               "(there is no CPU fallback)" % (what, self.flat.device))

    def state_dict_of(self, flat):
        """An OrderedDict view of a flat buffer laid out like this arena (another client's gathered state): fp32 entries only."""
        from collections import OrderedDict
        out, off = OrderedDict(), 0
        named = [(k, v) for k, v in self.model.named_parameters()] + [(k, b) for k, b in self.model.named_buffers() if b.dtype == torch.float32]
        for k, v in named:
            out[k] = flat[off:off + v.numel()].view(v.shape)
            off += (v.numel() + 3) // 4 * 4
        return out

    def zero_grad(self):
        nn_ops.fill_(self.grad) if self.grad.is_cuda else self.grad.zero_()
        for p, off in self._slots():
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)

    def _slots(self):
        off = 0
        for p in self.params:
            yield p, off
            off += (p.numel() + 3) // 4 * 4

    def key_ranges(self, pred):
        """Merged [start, end) element ranges of the arena holding the state entries whose state_dict key
        satisfies ``pred`` (layout: parameters in named_parameters() order, then the fp32 buffers)."""
        named = [(k, v) for k, v in self.model.named_parameters()]
        named += [(k, b) for k, b in self.model.named_buffers() if b.dtype == torch.float32]
        out, off = [], 0
        for k, v in named:
            n = (v.numel() + 3) // 4 * 4
            if pred(k):
                if out and out[-1][1] == off:
                    out[-1][1] = off + n
                else:
                    out.append([off, off + n])
            off += n
        assert off == self.n_total
        return [(a, b) for a, b in out]


class SGD(object):
    """optim.SGD(params, lr) with no momentum / weight decay (fed_run.py:657): p -= lr * g, one launch."""

    def __init__(self, model_or_params, lr):
        if isinstance(model_or_params, nn.Module):
            self.model = model_or_params
        elif isinstance(model_or_params, FlatParams):
            self.model = model_or_params.model
        else:
            raise TypeError("ccst_amd.fed.SGD takes the model (or its FlatParams), e.g. SGD(model, lr=args.lr)")
        self._arena = FlatParams.of(self.model)
        self.lr = float(lr)
        self.param_groups = [{"lr": self.lr, "params": self._arena.params}]

    @property
    def arena(self):
        """The optimiser is created before train() moves the model (fed_run.py:657 then :32): re-resolve the arena when the
        model's tensors have moved since (cheap pointer check per call, full check + rebuild only on a mismatch)."""
        a = self._arena
        if self.model.__dict__.get("_ccst_arena") is not a or not a.valid():
            a = self._arena = FlatParams.of(self.model)
            self.param_groups[0]["params"] = a.params
        return a

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def step(self, prepack=True):
        """prepack=False: the caller refreshes the packed weights itself (a captured train step does so at the HEAD of the next replay,
        see _GraphedTrainStep)."""
        a = self.arena
        a.require_cuda("SGD.step")
        check(_lib.load().ccst_sgd_f32(ptr(a.flat), ptr(a.grad), float(self.param_groups[0]["lr"]), a.n_param, stream_ptr()), "sgd")
        ops.bump_weights_epoch()
        if prepack:
            nn_ops.prepack_on_side(a.model)


_ONES = {}


def backward(loss):
    """loss.backward() with a cached root gradient of one: autograd otherwise materialises ones_like(loss) with an ATen fill
    kernel on every step."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones(loss.shape, device=loss.device, dtype=loss.dtype)
    loss.backward(one)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() (mean).  The last call's argmax==label count is kept in ``correct``
    (a device int32), so train()/test() need no second pass over the logits."""

    def __init__(self):
        super().__init__()
        self.correct = None

    def forward(self, logits, labels):
        if self.correct is None or self.correct.device != logits.device:
            self.correct = torch.zeros(1, device=logits.device, dtype=torch.int32)
        return nn_ops.CrossEntropyFn.apply(logits, labels, self.correct)


# ---------------------------------------------------------------------------
# train / test  (fed_run.py:31-88, :214-259)
# ---------------------------------------------------------------------------
class StepWindow(object):
    """Bounds how far the host runs ahead of the GPU in a train loop: tick() at the end of every iteration records an event and
    waits for the one `depth` iterations back.  Nothing in the loop synchronises otherwise (the reference's two .item() per
    iteration, fed_run.py:69,76, are accumulated on the device), and a host that issues a 12 ms step while the GPU runs it in 20 ms
    gets arbitrarily far ahead: every block that was last used on the weight-gradient stream stays unusable until its event has
    completed ON THE GPU, so the caching allocator keeps hipMalloc-ing instead of reusing (measured: 11 -> 40 GB and 440 device
    allocations in 25 ResNet50 steps).  With a window of 2 the GPU never waits for the host and the working set stays put."""

    def __init__(self, depth=2):
        self.depth, self.events = depth, []

    def tick(self):
        e = torch.cuda.Event()
        e.record()
        self.events.append(e)
        if len(self.events) > self.depth:
            self.events.pop(0).synchronize()


def _dg(args):
    return (getattr(args, "dg_method", "") or "").lower()


class _GraphedTrainStep(object):
    """One train iteration captured into a HIP graph and replayed per batch -- ROTATED: the replay starts with the re-pack of
    the weights the PREVIOUS step's SGD wrote (forked to the side stream, joined where the forward first reads a packed weight:
    it runs under zero_grad / stem conv / stem BatchNorm + pool exactly as in the eager loop), then zero_grad, forward, loss,
    running loss / accuracy sums, backward, SGD step.  A capture must end with every forked stream re-joined; with the re-pack
    at the TAIL of the step (round 5) that join serialised it behind the optimiser step -- 16.3 ms replayed against 14.9 ms
    of eager device time for ResNet50 at B=64 (BENCH_r05).  Weights rewritten outside the graph (FedAvg, load_state_dict, an
    eager step) are picked up by the next replay's own re-pack.  Results are bit-identical to the eager loop
    (tests/test_resnet_gpu.py::test_train_hip_graph_matches_eager).  Opt-in: args.hip_graph / --hip_graph / CCST_TRAIN_GRAPH=1,
    or "auto" (train() times both loops and keeps the faster)."""

    def __init__(self, model, optimizer, loss_fun, img, class_l, loss_all, correct_all):
        self.model, self.shape = model, (tuple(img.shape), tuple(class_l.shape))
        self.x, self.y = img.clone(), class_l.clone()
        dev = img.device
        # the epoch's running sums: the SAME tensors the eager iterations add to, so that the loss is summed in iteration order
        # whatever mix of eager and replayed iterations an epoch has (separate sums added at the end differ in the last bit)
        self.loss_all, self.correct_all = loss_all, correct_all
        self.convs = [m for m in model.modules() if hasattr(m, "prepack")]
        self.graph = torch.cuda.CUDAGraph()
        ops.reset_absmax_pool()                 # the step's |max| word rows: from a block zero-filled INSIDE the graph
        nn_ops.reset_deferred()
        with torch.cuda.graph(self.graph):
            nn_ops.prepack_on_side(model)       # head of the replay: the packs of the weights as they are NOW (joined in the forward)
            optimizer.zero_grad()
            loss = loss_fun(model(self.x), self.y)
            self.loss_all += loss.detach()
            self.correct_all += loss_fun.correct[0]
            backward(loss)
            optimizer.step(prepack=False)
            nn_ops.join_prepack(dev)            # (a model without packed convs never joined in its forward)
        ops.reset_absmax_pool()
        nn_ops.restamp_packs(model)             # host-side keys: the packs are as current as the next replay needs them

    def run(self, img, class_l):
        self.x.copy_(img, non_blocking=True)
        self.y.copy_(class_l, non_blocking=True)
        for m in self.convs:        # weights rewritten outside the graph (FedAvg, load_state_dict, an eager step):
            m.prepack()             # re-pack now; otherwise this is a key compare
        self.graph.replay()
        # A replay updates the weights and their packed copies on the device without touching the host-side cache
        # keys; the caller bumps ops.WEIGHTS_EPOCH before anything eager reads a packed weight again (train()).


def _graph_mode(args):
    """'on' (args.hip_graph true / --hip_graph / CCST_TRAIN_GRAPH=1), 'auto' (args.hip_graph == 'auto' / CCST_TRAIN_GRAPH=auto:
    train() times AUTO_PROBE iterations of the eager loop and of the replayed one per batch shape and keeps the faster -- both give
    the same bits, so the choice is free), else 'off'."""
    import os
    v = getattr(args, "hip_graph", None)
    if v is None or v is False:
        v = os.environ.get("CCST_TRAIN_GRAPH", "0")
    if isinstance(v, str):
        v = v.strip().lower()
        return "auto" if v == "auto" else ("on" if v in ("1", "true", "yes", "on") else "off")
    return "on" if v else "off"


def _graph_wanted(args):
    return _graph_mode(args) != "off"


AUTO_PROBE = 10
AUTO_EAGER_MARGIN = 1.03


class _LoopTimer(object):
    """Times P consecutive iterations of one loop form with two events on the compute stream (they see host-bound gaps as well as
    device time): `before_iteration()` is called ahead of each iteration and returns ms per iteration once, after the P-th."""

    def __init__(self, P, skip=0):
        self.P, self.n, self.e0, self.skip = P, 0, None, skip

    def before_iteration(self):
        if self.skip > 0:
            self.skip -= 1
            return None
        if self.e0 is None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
            self.n = 0
            return None
        self.n += 1
        if self.n < self.P:
            return None
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        e1.synchronize()
        ms, self.e0 = self.e0.elapsed_time(e1) / self.P, None
        return ms


def train(model, train_loader, optimizer, loss_fun, client_num, device, args, iter_idx, logger):
    if _dg(args) in ("rsc", "jigsaw", "mixstyle"):
        raise NotImplementedError("ccst_amd.fed: --dg_method %s is outside the hot path" % args.dg_method)
    model.to(device)
    model.train()
    num_data = 0
    loss_all = torch.zeros((), device=device)
    correct_all = torch.zeros((), device=device, dtype=torch.int64)
    it = -1
    fused_acc = isinstance(loss_fun, CrossEntropyLoss)
    use_graph = _graph_wanted(args) and fused_acc and logger is None and isinstance(optimizer, SGD)
    auto = use_graph and _graph_mode(args) == "auto"
    if use_graph:
        # Captured steps hold raw addresses of the parameter arena.  model.to() above (args.offload_models moves the model to the
        # CPU and back every epoch), a load-by-assignment or any model.cpu() by the caller re-homes the tensors: re-resolve the
        # arena FIRST -- FlatParams.of() rebuilds it when the tensors moved and drops the captured steps with it -- instead of
        # replaying a graph that reads and writes freed memory and silently stops training the live tensors.
        FlatParams.of(model)
    steps = model.__dict__.setdefault("_ccst_graph_steps", {}) if use_graph else None
    if use_graph:           # captured steps add into these (static addresses): one pair per model, zeroed per epoch
        acc = steps.get("_sums")
        if acc is None:
            acc = steps["_sums"] = (loss_all, correct_all)
        loss_all, correct_all = acc
        loss_all.zero_()
        correct_all.zero_()
    eager_iters, stale_keys = 0, False
    window = StepWindow() if torch.device(device).type == "cuda" else None
    timers = {}             # auto: per batch shape, the timer of the loop form being measured in THIS call (a measurement never spans calls)
    for it, data in enumerate(train_loader):
        img, class_l = data
        img, class_l = img.to(device, non_blocking=True), class_l.to(device, non_blocking=True)
        num_data += img.size(0)
        if use_graph:
            key = (tuple(img.shape), tuple(class_l.shape), float(optimizer.lr), id(loss_fun))
            gs = steps.get(key)
            st = steps.setdefault("_auto", {}).setdefault(key, {"eager_ms": None, "graph_ms": None}) if auto else None
            ready = True
            if st is not None:
                if st["eager_ms"] is None:              # phase 0: time the eager loop first (these ARE training iterations)
                    ready = False
                    if eager_iters >= 2:
                        st["eager_ms"] = timers.setdefault((key, 0), _LoopTimer(AUTO_PROBE)).before_iteration()
                        ready = st["eager_ms"] is not None
                elif gs is not None and st["graph_ms"] is None:         # phase 1: ... then the replayed one (not its first replay)
                    st["graph_ms"] = timers.setdefault((key, 1), _LoopTimer(AUTO_PROBE, skip=1)).before_iteration()
                if st["graph_ms"] is not None and st["graph_ms"] > st["eager_ms"] * AUTO_EAGER_MARGIN:
                    gs, ready = None, False             # decided: the eager loop is the faster one for this shape (by a margin: its
                                                        # time depends on the host, the replay's does not)
            if gs is None and ready and eager_iters >= 2 and len([k for k in steps if k not in ("_sums", "_auto")]) < 4:     # capture once caches / workspaces are warm (<= 4 shapes)
                nn_ops.join_prepack(device)
                torch.cuda.current_stream(device).synchronize()
                gs = steps[key] = _GraphedTrainStep(model, optimizer, loss_fun, img, class_l, loss_all, correct_all)
            if gs is not None:
                gs.run(img, class_l)
                stale_keys = True
                if window is not None:
                    window.tick()
                continue
            if stale_keys:          # an eager step after replays: the host-side packed-weight keys are behind the device
                ops.bump_weights_epoch()
                stale_keys = False
        eager_iters += 1
        optimizer.zero_grad()
        class_logit = model(img)
        loss = loss_fun(class_logit, class_l)
        if fused_acc:
            batch_correct = loss_fun.correct[0]
        else:
            batch_correct = torch.sum(class_logit.max(dim=1)[1] == class_l.data)
        # the reference syncs twice per iteration on .item() (fed_run.py:69,76); accumulate on device instead
        loss_all += loss.detach()
        correct_all += batch_correct
        if logger is not None:
            logger.log(it, len(train_loader), {"train_loss": loss.item()}, {"class_acc": int(batch_correct)}, img.shape[0])
        backward(loss)
        optimizer.step()
        del img, class_l
        if window is not None:
            window.tick()
    if use_graph and stale_keys:
        ops.bump_weights_epoch()
    train_loss = float(loss_all) / (it + 1)
    train_acc = float(correct_all) / num_data
    # fed_run.py:85 moves the model back to the CPU after every client epoch; on MI355X the client
    # model stays resident in HBM (288 GB): pass args.offload_models=True to reproduce the move.
    if getattr(args, "offload_models", False):
        model.to('cpu')
    return train_loss, train_acc


def test(model, test_loader, loss_fun, device, args):
    if getattr(args, "IN_test", False):
        raise NotImplementedError("ccst_amd.fed: --IN_test is outside the hot path")
    model.to(device)
    model.eval()
    num_data = 0
    loss_all = torch.zeros((), device=device)
    correct_all = torch.zeros((), device=device, dtype=torch.int64)
    it = -1
    fused_acc = isinstance(loss_fun, CrossEntropyLoss)
    with torch.no_grad():
        for it, (data, class_l) in enumerate(test_loader):
            data, class_l = data.to(device, non_blocking=True), class_l.to(device, non_blocking=True)
            class_logit = model(data)
            loss = loss_fun(class_logit, class_l)
            loss_all += loss
            correct_all += loss_fun.correct[0] if fused_acc else torch.sum(class_logit.max(dim=1)[1] == class_l.data)
            num_data += data.size(0)
    class_acc = float(correct_all) / num_data
    test_loss = float(loss_all) / (it + 1)
    if getattr(args, "offload_models", False):
        model.to('cpu')
    return test_loss, class_acc


# ---------------------------------------------------------------------------
# FedAvg
# ---------------------------------------------------------------------------
def _axpy(dst, src, w):
    """dst += w * src on flat fp32 tensors (16-byte aligned), HIP."""
    check(_lib.load().ccst_sgd_f32(ptr(dst), ptr(src), -float(w), dst.numel(), stream_ptr()), "axpy")


def _copy_counters(server_model, client0):
    """server.num_batches_tracked <- client 0's, for every BatchNorm (fed_run.py:404-405): one copy of the int64 counter arena where both
    models keep one (nets/resnet.ResNet.counter_arena), else key by key."""
    sa = server_model.counter_arena() if hasattr(server_model, "counter_arena") else None
    ca = client0.counter_arena() if hasattr(client0, "counter_arena") else None
    if sa is not None and ca is not None and sa.numel() == ca.numel() and sa.device == ca.device:
        sa.copy_(ca)
        return
    ssd, c0 = server_model.state_dict(), client0.state_dict()
    for key in ssd.keys():
        if 'num_batches_tracked' in key:
            ssd[key].data.copy_(c0[key])


def communication(args, server_model, models, client_weights):
    """fed_run.py:385-455, the branch every non-'fedbn' mode takes (:400-414): for each state key,
    'num_batches_tracked' -> server takes client 0's value (clients keep theirs); otherwise
    server = sum_i w_i * client_i and every client is overwritten with it.  In-process form (all
    models on one GPU) for API parity; see communication_distributed for the one-client-per-GPU form."""
    mode = (getattr(args, "mode", "fedavg") or "fedavg").lower()
    with torch.no_grad():
        dev = next(models[0].parameters()).device
        for m in [server_model] + list(models):          # the reference averages on the CPU after train() moved the clients back
            if next(m.parameters()).device != dev:       # (fed_run.py:85); here everything meets on the clients' GPU
                m.to(dev)
        arenas = [FlatParams.of(m) for m in [server_model] + list(models)]
        arenas[0].require_cuda("communication()")
        nn_ops.join_prepack(arenas[0].flat.device)      # side-stream re-packs still read the client weights
        srv, clients = arenas[0], arenas[1:]
        if any(c.n_total != srv.n_total for c in clients):
            raise ValueError("communication: client and server models differ in size")
        K = len(client_weights)
        fused = mode != "fedbn" and 1 <= K <= 16 and len({c.flat.data_ptr() for c in clients[:K]} | {srv.flat.data_ptr()}) == K + 1
        if fused:
            # one pass: K reads, K + 1 writes per element (ccst_fedavg_f32; the same rounding sequence as the loop below)
            import ctypes
            cl = (ctypes.c_void_p * K)(*[c.flat.data_ptr() for c in clients[:K]])
            cw = (ctypes.c_float * K)(*[float(w) for w in client_weights])
            check(_lib.load().ccst_fedavg_f32(ptr(srv.flat), cl, cw, K, srv.flat.numel(), stream_ptr()), "fedavg")
        else:
            nn_ops.fill_(srv.flat, 0.0)
            for ci in range(K):
                _axpy(srv.flat, clients[ci].flat, client_weights[ci])
        if fused:
            pass
        elif mode == "fedbn":
            # fed_run.py:388-399: the server averages everything, clients keep every key whose name contains 'bn'
            shared = srv.key_ranges(lambda k: 'bn' not in k)
            for ci in range(len(client_weights)):
                for a, b in shared:
                    clients[ci].flat[a:b].copy_(srv.flat[a:b])
        else:
            for ci in range(len(client_weights)):
                clients[ci].flat.copy_(srv.flat)                  # device-to-device memcpy
        ops.bump_weights_epoch()
        _copy_counters(server_model, models[0])
    return server_model, models


def test_fedbn(server_model, models, test_loader, loss_fun, device, args):
    """fed_run.py:350-381 (``--test`` of a ``--mode fedbn`` checkpoint): the server takes client 0's ``num_batches_tracked`` and, for
    every other entry whose key contains 'bn', the 1/K average of the clients' local entries; then an ordinary test() pass."""
    client_num = len(models)
    w = float(1. / client_num)
    with torch.no_grad():
        for m in [server_model] + list(models):
            m.to(device)
        srv = FlatParams.of(server_model)
        srv.require_cuda("test_fedbn()")
        nn_ops.join_prepack(srv.flat.device)
        clients = [FlatParams.of(m) for m in models]
        for a, b in srv.key_ranges(lambda k: 'bn' in k):
            srv.flat[a:b].zero_()
            for c in clients:
                _axpy(srv.flat[a:b], c.flat[a:b], w)
        ops.bump_weights_epoch()
        ssd, c0 = server_model.state_dict(), models[0].state_dict()
        for key in ssd.keys():
            if 'num_batches_tracked' in key:
                ssd[key].data.copy_(c0[key])
    return test(server_model, test_loader, loss_fun, device, args)


def gather_client_states(model, group=None, dst=0):
    """One client per rank: rank `dst` receives every client's full state (flat fp32 arena over RCCL + the int64 counters) and
    returns [state_dict_0, ..., state_dict_{K-1}] on the CPU -- what the fedbn checkpoint stores as 'model_{k}' (fed_run.py:735-739);
    other ranks return None."""
    import torch.distributed as dist
    from collections import OrderedDict
    arena = FlatParams.of(model)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nn_ops.join_prepack(arena.flat.device) if arena.flat.is_cuda else None
    keys = list(model.state_dict().keys())
    ints = [v for k, v in model.state_dict().items() if v.dtype != torch.float32]
    cnt = torch.stack([v.reshape(()).to(arena.flat.device, torch.int64) for v in ints]) if ints else torch.zeros(0, dtype=torch.int64, device=arena.flat.device)
    flats = [torch.empty_like(arena.flat) for _ in range(world)] if rank == dst else None
    cnts = [torch.empty_like(cnt) for _ in range(world)] if rank == dst else None
    dist.gather(arena.flat, flats, dst=dst, group=group)
    dist.gather(cnt, cnts, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for k in range(world):
        fl = arena.state_dict_of(flats[k].cpu())
        it = iter(cnts[k].cpu())
        sd = OrderedDict()
        for key in keys:
            sd[key] = fl[key].clone() if key in fl else next(it).clone()
        out.append(sd)
    return out


def _hip_scale(flat, w, n):
    if not flat.is_cuda:
        raise RuntimeError("ccst_amd.fed: the FedAvg pre-scale runs on the GPU; no CPU fallback")
    check(_lib.load().ccst_scale_f32(ptr(flat), float(w), n, stream_ptr()), "scale")


def communication_distributed(args, model, client_weight, group=None, server_counters=None, scale_fn=_hip_scale,
                              server_model=None):
    """FedAvg with one client per rank/GPU: theta <- sum_i w_i * theta_i as ONE all-reduce(SUM) of the
    flat arena (parameters + BN running stats) over RCCL/xGMI; each rank pre-scales by its own w_i.
    After the call every rank holds the server model (what fed_run.py:411-414 copies to all clients).
    num_batches_tracked: each client keeps its own; the server's copy is client 0's (fed_run.py:404-405),
    returned in `server_counters` (broadcast from rank 0) when given.
    --mode fedbn (fed_run.py:388-399): the client keeps every entry whose key contains 'bn', so the averaged
    model needs a home of its own: pass ``server_model`` (a replica on this rank); it receives the full
    average and the client only the shared (non-'bn') entries."""
    import torch.distributed as dist
    mode = (getattr(args, "mode", "fedavg") or "fedavg").lower()
    arena = FlatParams.of(model)
    with torch.no_grad():
        if arena.flat.is_cuda:
            nn_ops.join_prepack(arena.flat.device)           # side-stream re-packs still read these weights
        if mode == "fedbn":
            if server_model is None:
                raise ValueError("communication_distributed: --mode fedbn needs server_model (clients keep their BN entries)")
            srv = FlatParams.of(server_model)
            if srv.n_total != arena.n_total:
                raise ValueError("communication: client and server models differ in size")
            srv.flat.copy_(arena.flat)
            scale_fn(srv.flat, client_weight, srv.n_total)
            dist.all_reduce(srv.flat, op=dist.ReduceOp.SUM, group=group)
            for a, b in arena.key_ranges(lambda k: 'bn' not in k):
                arena.flat[a:b].copy_(srv.flat[a:b])
        else:
            scale_fn(arena.flat, client_weight, arena.n_total)      # tests of the gloo protocol inject a host scale
            dist.all_reduce(arena.flat, op=dist.ReduceOp.SUM, group=group)
            if server_model is not None and server_model is not model:
                FlatParams.of(server_model).flat.copy_(arena.flat)
        ops.bump_weights_epoch()
        if server_counters is not None:
            for t in server_counters:
                dist.broadcast(t, src=0, group=group)
    return model
