"""Executes ``graph`` nodes as fused HIP launches (the ``sess.run`` of the reference).

Fusion rules (all arithmetic stays in the kernels of libmpgan_hip.so):
  * conv2d -> bias_add -> [batch_norm(inference)] -> [activation] -> [pixel_norm]
    is one ``mpg_conv2d_fused`` launch; batch norm is folded into the packed
    weights and the bias (GAN.py:108-113);
  * ``add`` of two such linear chains (the residual shortcut of resBlock,
    multipassGAN-4x.py:523) becomes one launch with two K-segments;
  * a conv reading ``concat`` / nearest ``resize`` / channel ``slice`` nodes reads
    their sources directly (multipassGAN-out.py:357; GAN.py:517);
  * ``chain + tensor`` with a linear chain on one side is the epilogue post-add
    (addBicubicUpsample, multipassGAN-out.py:327-332).
Everything else falls back to one kernel per node.
"""
import re

import numpy as np
import torch

from . import _lib, ops
from . import graph as G


class VariableStore(object):
    """Device-resident parameters keyed by TF variable path (GAN.py:668,683)."""

    def __init__(self, device="cuda:0", seed=777, bn_seed=4321):
        self.device = torch.device(device)
        self.values = {}
        self.version = 0
        self.seed, self.bn_seed = seed, bn_seed

    @staticmethod
    def synthetic(name, shape, kind, seed=777, bn_seed=4321):
        """Synthetic init: weight ~ N(0,1) (GAN.py:668), bias 0.1 (GAN.py:683),
        batch-norm statistics perturbed so BN is not the identity.  Name-keyed streams."""
        h = 0
        for ch in name:
            h = (h * 131 + ord(ch)) % (2 ** 31 - 1)
        if kind == "weight":
            return np.random.default_rng([seed, h]).standard_normal(shape).astype(np.float32)
        if kind == "bias":
            return np.full(shape, 0.1, dtype=np.float32)
        rng = np.random.default_rng([bn_seed, h])
        if kind == "gamma":
            return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        if kind in ("beta", "moving_mean"):
            return (0.1 * rng.standard_normal(shape)).astype(np.float32)
        if kind == "moving_variance":
            return (1.0 + 0.2 * rng.random(shape)).astype(np.float32)
        raise G.GraphError("unknown variable kind %r" % (kind,))

    def _place(self, t):
        # without a GPU (graph building / plan inspection on the build host) values stay on the CPU
        if self.device.type == "cuda" and not torch.cuda.is_available():
            return t
        return t.to(self.device)

    def ensure(self, graph):
        for name in [k for k, v in self.values.items() if v.device != self.device]:
            self.values[name] = self._place(self.values[name])
        for name, spec in graph.variables.items():
            if name not in self.values:
                self.set(name, self.synthetic(name, spec.shape, spec.kind, self.seed, self.bn_seed))
            elif tuple(self.values[name].shape) != spec.shape:
                raise G.GraphError("variable %s has shape %s, graph expects %s"
                                   % (name, tuple(self.values[name].shape), spec.shape))

    def set(self, name, value):
        self.values[name] = self._place(torch.as_tensor(np.ascontiguousarray(value), dtype=torch.float32))
        self.version += 1

    _SLOT_KEY = re.compile(r"(^|/)(Adam(_\d+)?|beta[12]_power(_\d+)?|adam_t|ls_var|ExponentialMovingAverage)$")

    def load(self, params, prefix=""):
        """model variables of a checkpoint; optimiser slots (TF Saver names) are the trainer's to restore"""
        for k, v in params.items():
            if not self._SLOT_KEY.search(k):
                self.set(prefix + k, v)

    def numpy(self):
        return {k: v.cpu().numpy() for k, v in self.values.items()}

    def get(self, name):
        return self.values[name]


class _Term(object):
    """bn(bias_add(conv2d(src, W))) -- one linear term of a fused convolution."""

    __slots__ = ("conv", "bias", "bn")

    def __init__(self, conv, bias=None, bn=None):
        self.conv, self.bias, self.bn = conv, bias, bn


def _is_pow2(v):
    return v >= 1 and (v & (v - 1)) == 0


class _Plan(object):
    def __init__(self):
        self.steps = []       # (node, callable(env) -> tensor)
        self.last_use = {}    # node id -> index of the last step reading it
        self.free_after = []  # per step: node ids whose tensors are dead afterwards


class Session(object):
    def __init__(self, device="cuda:0", prec=ops.DEFAULT_PREC, variables=None, graph=None, prec_map=None):
        self.device = torch.device(device)
        self.prec = prec
        # per-launch precision override: [(substring of the first term's weight name, prec), ...]
        self.prec_map = list(prec_map or [])
        self.graph = graph or G.get_default_graph()
        self.vars = variables or VariableStore(device)
        self._plans = {}
        self._packed = {}
        self._folded = {}
        self._cache_version = -1
        # measurement hook: when `tap` is a substring of a fused launch's leading weight name, the launch's
        # operands (G8 segments taken from the running pipeline, bias, epilogue flags) are kept in `tapped`
        # so that bench.py can re-issue exactly that launch on exactly those activations
        self.tap = None
        self.tapped = None
        self.tap_events = None     # a list: every tapped launch is bracketed by a (start, end) pair of timing events on its stream

    # ------------------------------------------------------------------ public
    def run(self, fetches, feed_dict=None):
        """numpy in / numpy out, like ``sess.run(sampler, feed_dict={x: ...})``."""
        single = isinstance(fetches, G.Node)
        flist = [fetches] if single else list(fetches)
        feeds = {}
        self._scalars = {}
        for k, v in (feed_dict or {}).items():
            if isinstance(k, G.Scalar):
                self._scalars[k.node] = float(v)          # e.g. `percentage` of the growing nets
            elif isinstance(k, G.Node):
                feeds[k] = torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float32).to(self.device)
        outs = [self.run_device(f, feeds).cpu().numpy() for f in flist]
        return outs[0] if single else outs

    def run_device(self, fetch, feeds, out=None):
        """device tensors in / device tensor out (what the multi-pass pipeline uses).  out: a contiguous fp32 buffer with
        as many elements as the result; the launch that produces the fetched value writes it there when it is a fused
        convolution (the slice batches of a pass then land in one volume without a concatenation pass)."""
        _lib.load()
        if not torch.cuda.is_available():
            raise _lib.MpgError("no GPU visible: the multi-pass GAN path has no CPU fallback")
        self.vars.ensure(self.graph)
        if self._cache_version != self.vars.version:
            self._packed.clear()
            self._folded.clear()
            self._cache_version = self.vars.version
        plan = self._plans.get(fetch.id)
        if plan is None:
            plan = self._compile(fetch)
            self._plans[fetch.id] = plan
        env = {}
        for node, t in feeds.items():
            env[node.id] = t
        producer = self._producer_of(fetch) if out is not None else None
        for i, (node, fn) in enumerate(plan.steps):
            env["__out__"] = out if (producer is not None and node.id == producer.id) else None
            env[node.id] = fn(env)
            for nid in plan.free_after[i]:
                env.pop(nid, None)
        res = self._f32(env, fetch)
        if out is not None and res.data_ptr() != out.data_ptr():
            out.view(-1).copy_(res.reshape(-1))
            res = out.view(res.shape)
        return res

    @staticmethod
    def _producer_of(node):
        """the node whose buffer `node` aliases: through reshapes"""
        while node.op == "reshape":
            node = node.inputs[0]
        return node

    # ------------------------------------------------------------------ tensor formats
    # A fused convolution can emit fp32 NHWC and / or the G8 layout (in either flavour) its consumers
    # DMA from; its env entry is a dict {"f32": tensor | None, "g8": {flavour: G8}}.  Every other
    # node holds an fp32 tensor.
    @staticmethod
    def _f32(env, node):
        v = env[node.id]
        if isinstance(v, dict):
            if v["f32"] is None:
                v["f32"] = ops.from_g8(v["g8"][ops.G8_F16])
            return v["f32"]
        return v

    @staticmethod
    def _g8(env, node, c_off, cin, flavour):
        """(G8 tensor, channel offset inside it) holding channels [c_off, c_off+cin) of `node`"""
        v = env[node.id]
        if isinstance(v, dict) and flavour in v["g8"] and c_off % 8 == 0:
            return v["g8"][flavour], c_off
        key = ("g8", node.id, c_off, cin, flavour)
        g = env.get(key)
        if g is None:
            g = ops.to_g8(Session._f32(env, node), c_off, cin, flavour)
            env[key] = g
        return g, 0

    # ------------------------------------------------------------------ compile
    def _compile(self, fetch):
        consumers = {}
        order = []
        seen = set()

        def visit(n):
            if n.id in seen:
                return
            seen.add(n.id)
            for i in n.inputs:
                consumers.setdefault(i.id, []).append(n)
                visit(i)
            order.append(n)

        visit(fetch)
        self._consumers = consumers
        self._fetch = fetch
        plan = _Plan()
        done = set()
        fused_steps = []       # (node, run) of every fused convolution
        need_f32 = {fetch.id}  # nodes somebody reads as fp32 NHWC
        need_g8 = set()        # (node id, flavour) of fused outputs read by another fused convolution

        def single_use(n):
            return len(consumers.get(n.id, [])) == 1 and n is not fetch

        def emit(n):
            if n.id in done:
                return
            done.add(n.id)
            if n.op == "variable":
                plan.steps.append((n, lambda env, nm=n.attrs["var"]: self.vars.get(nm)))
                return
            if n.op == "placeholder":
                def feed(env, node=n):
                    if node.id not in env:
                        raise G.GraphError("placeholder %s was not fed" % node.name)
                    return env[node.id]
                plan.steps.append((n, feed))
                return
            fused = self._match_fused(n, single_use)
            if fused is not None:
                deps, fn = fused
                fused_steps.append((n, fn))
            else:
                deps, fn = self._fallback(n, single_use)
                for d in deps:
                    need_f32.add(d.id)
            for d in deps:
                emit(d)
            idx = len(plan.steps)
            plan.steps.append((n, fn))
            for d in deps:
                plan.last_use[d.id] = idx

        emit(fetch)
        self._fuse_small_pairs(plan, fused_steps, consumers, fetch)
        fused_steps = [(n, fn) for n, fn in fused_steps if fn.info.get("kind") != "fused_into_next"]
        fused_ids = set(n.id for n, _ in fused_steps)
        for n, fn in fused_steps:
            fl = ops.flavour_for(fn.info["prec"])
            for seg in fn.info["segments"]:
                if seg["src_id"] in fused_ids and seg["c_off"] % 8 == 0:
                    need_g8.add((seg["src_id"], fl))
                else:
                    need_f32.add(seg["src_id"])
            if fn.info["post_add_id"] is not None:
                need_f32.add(fn.info["post_add_id"])
        for n, fn in fused_steps:
            fn.emit["g8"] = (n.id, ops.G8_F16) in need_g8
            fn.emit["f32"] = n.id in need_f32 or not fn.emit["g8"]
        # variables, placeholders and the fetch stay alive; everything else dies after its last reader
        plan.free_after = [[] for _ in plan.steps]
        keep = set(n.id for n, _ in plan.steps if n.op in ("variable", "placeholder"))
        keep.add(fetch.id)
        for nid, idx in plan.last_use.items():
            if nid not in keep:
                plan.free_after[idx].append(nid)
        return plan

    # ---- residual blocks of small-channel convolutions: two launches -> one -------------------------------
    def _fuse_small_pairs(self, plan, fused_steps, consumers, fetch):
        """relu(convB(relu(convA(x))) + conv1x1(x)) with <= 8 channels everywhere (resBlock 0 and 3 of gen_resnet,
        multipassGAN-4x.py:505-526,560,564) was planned as two conv_small launches with the middle tensor going through
        HBM; here launch A is dropped and launch B replaced by one mpg_conv2d_small_pair call."""
        by_id = dict((n.id, (n, fn)) for n, fn in fused_steps)
        index = dict((n.id, i) for i, (n, _) in enumerate(plan.steps))
        for n2, fn2 in list(fused_steps):
            p2 = getattr(fn2, "parts", None)
            if p2 is None or p2["pn"] or p2["post_add"] is not None or p2["cout"] > 8 or not 1 <= len(p2["segs"]) <= 2:
                continue
            src1, c_off1, up1, term_b, w_off_b, cin_b = p2["segs"][0]
            if src1.id not in by_id or c_off1 != 0 or up1 != 0 or w_off_b != 0:
                continue
            n1, fn1 = by_id[src1.id]
            p1 = getattr(fn1, "parts", None)
            if (p1 is None or n1 is fetch or len(consumers.get(n1.id, [])) != 1 or p1["pn"] or p1["post_add"] is not None
                    or len(p1["segs"]) != 1 or p1["prec"] != p2["prec"] or p1["cout"] != cin_b):
                continue
            src0, c_off0, up0, term_a, w_off_a, cin_a = p1["segs"][0]
            if w_off_a != 0 or cin_a != term_a.conv.inputs[1].shape[2] or cin_b != term_b.conv.inputs[1].shape[2]:
                continue
            term_s = None
            if len(p2["segs"]) == 2:
                src_s, c_off_s, up_s, term_s, w_off_s, cin_s = p2["segs"][1]
                if src_s is not src0 or c_off_s != c_off0 or up_s != up0 or w_off_s != 0 or cin_s != cin_a:
                    continue
            ka, kb = tuple(term_a.conv.inputs[1].shape[:2]), tuple(term_b.conv.inputs[1].shape[:2])
            ks = tuple(term_s.conv.inputs[1].shape[:2]) if term_s is not None else None
            if not ops.small_pair_ok(cin_a, p1["cout"], p2["cout"], ka, kb, ks, planner=True):
                continue
            emit2 = fn2.emit

            def run(env, src0=src0, c_off0=c_off0, up0=up0, cin_a=cin_a, term_a=term_a, term_b=term_b, term_s=term_s,
                    p1=p1, p2=p2, emit2=emit2):
                prec = p2["prec"]
                pk_a = self._packed_for(term_a, 0, cin_a, prec)
                pk_b = self._packed_for(term_b, 0, p1["cout"], prec)
                pk_s = self._packed_for(term_s, 0, cin_a, prec) if term_s is not None else None
                g8, off = self._g8(env, src0, c_off0, cin_a, ops.G8_F16)
                res = ops.conv2d_small_pair(g8, off, up0, pk_a, pk_b, pk_s, p2["out_hw"], bias_a=self._bias_for(p1["terms"]),
                                            act_a=p1["act"], leak_a=p1["leak"], bias_b=self._bias_for(p2["terms"]),
                                            act_b=p2["act"], leak_b=p2["leak"], want_f32=emit2["f32"], want_g8=emit2["g8"],
                                            out=env.get("__out__") if emit2["f32"] else None)
                res = list(res) if isinstance(res, tuple) else [res]
                out = {"f32": None, "g8": {}}
                if emit2["f32"]:
                    out["f32"] = res.pop(0)
                if emit2["g8"]:
                    out["g8"][ops.G8_F16] = res.pop(0)
                return out

            run.emit = emit2
            run.parts = None
            run.info = dict(fn2.info)
            run.info["kind"] = "conv2d_small_pair"
            run.info["cmid"] = p1["cout"]
            run.info["segments"] = [dict(fn1.info["segments"][0], role="conv_a")] + [
                dict(sg, role="shortcut") for sg in fn2.info["segments"][1:]]
            run.info["act_a"] = p1["act"]

            def skipped(env):
                return None
            skipped.info = {"kind": "fused_into_next"}
            plan.steps[index[n1.id]] = (n1, skipped)
            plan.steps[index[n2.id]] = (n2, run)
            # the block input is now read by the second launch
            plan.last_use[src0.id] = max(plan.last_use.get(src0.id, 0), index[n2.id])
            fused_steps[fused_steps.index((n1, fn1))] = (n1, skipped)
            fused_steps[fused_steps.index((n2, fn2))] = (n2, run)
            del by_id[n1.id]

    # ---- pattern matching -------------------------------------------------
    def _match_term(self, n, single_use):
        """bn?(bias_add?(conv2d)) with stride 1, cout <= 128, k <= 7.  The root n may have any
        number of consumers (the caller decides); every inner node must feed this chain only."""
        bn = bias = None
        cur = n
        if cur.op == "batch_norm":
            if cur.attrs["training"]:
                return None
            bn = cur
            cur = cur.inputs[0]
            if not single_use(cur):
                return None
        if cur.op == "bias_add":
            bias = cur
            cur = cur.inputs[0]
            if not single_use(cur):
                return None
        if cur.op != "conv2d" or cur.attrs["stride"] != (1, 1):
            return None
        kh, kw, cin, cout = cur.inputs[1].shape
        if cout > 128 or kh > 7 or kw > 7 or cur.inputs[1].op != "variable":
            return None
        return _Term(cur, bias, bn)

    def _match_lin(self, n, single_use, top=True):
        if n.op == "add":
            if not top and not single_use(n):
                return None
            a = self._match_lin(n.inputs[0], single_use, False) if single_use(n.inputs[0]) else None
            b = self._match_lin(n.inputs[1], single_use, False) if single_use(n.inputs[1]) else None
            if a is not None and b is not None:
                return a + b
            return None
        t = self._match_term(n, single_use)
        return [t] if t is not None else None

    def _match_fused(self, n, single_use):
        post_add = None
        cur = n
        # chain + tensor  (only valid as the last op: the epilogue adds after act / pixel norm)
        if cur.op == "add" and self._match_lin(cur, single_use) is None:
            for side in (0, 1):
                cand, other = cur.inputs[side], cur.inputs[1 - side]
                if single_use(cand) and self._match_chain(cand, single_use) is not None:
                    post_add = other
                    cur = cand
                    break
            if post_add is None:
                return None
        chain = self._match_chain(cur, single_use, top=(post_add is None))
        if chain is None:
            return None
        terms, act, leak, pn, pn_eps = chain
        cout = terms[0].conv.shape[3]
        if any(t.conv.shape[3] != cout for t in terms):
            return None
        segs = []
        for t in terms:
            s = self._segments_of(t)
            if s is None:
                return None
            segs.extend(s)
        if len(segs) > _lib.MAX_SEG:
            return None
        deps = [s[0] for s in segs] + ([post_add] if post_add is not None else [])
        out_hw = (n.shape[1], n.shape[2])

        lead = terms[0].conv.inputs[1].attrs["var"]
        prec = self.prec
        for pat, pr in self.prec_map:
            if pat in lead:
                prec = pr
        if prec == ops.PREC_F16F6 and not ops.f6_available(
                cout, [(sg[3].conv.inputs[1].shape[0], sg[3].conv.inputs[1].shape[1], sg[5]) for sg in segs]):
            prec = ops.PREC_F16X3      # shapes the F16F6 kernels do not cover keep the fp16 split
        # short contractions on wide outputs (8 -> 128 5x5 of resBlock 1: K = 200, four cout tiles): a launch of 4 weight
        # stages per tile is all prologue, barriers and conversions; the three-product fp16 kernel (8-KB stages, no
        # conversions) is faster there AND fp32-grade (109 against 124 us per 8 slices, profiles/r03/kloop_variants.md)
        total_k = sum(sg[3].conv.inputs[1].shape[0] * sg[3].conv.inputs[1].shape[1] * sg[5] for sg in segs)
        if prec == ops.PREC_F16F6 and total_k <= ops.F16F6_MIN_K and cout > 8:
            prec = ops.PREC_F16X3

        emit = {"f32": True, "g8": False}

        def run(env, segs=segs, terms=terms, prec=prec):
            seg_objs = []
            for (src, c_off_src, up, term, w_off, cin) in segs:
                pk = self._packed_for(term, w_off, cin, prec)
                g8, off = self._g8(env, src, c_off_src, cin, ops.flavour_for(prec))
                seg_objs.append(ops.Segment(g8, pk, off, up))
            bias = self._bias_for(terms)
            pa = self._f32(env, post_add) if post_add is not None else None
            if self.tap is not None and self.tap in lead:
                self.tapped = dict(segments=seg_objs, out_hw=out_hw, bias=bias, act=act, leak=leak, pixel_norm=pn,
                                   pn_eps=pn_eps, post_add=pa, want_f32=emit["f32"], want_g8=emit["g8"])
            timed = self.tap is not None and self.tap_events is not None and self.tap in lead
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            dst = env.get("__out__") if emit["f32"] else None
            if dst is not None:
                dst = dst.view(seg_objs[0].x.n, out_hw[0], out_hw[1], cout)
            res = ops.conv2d_fused(seg_objs, out_hw, bias=bias, act=act, leak=leak, pixel_norm=pn, pn_eps=pn_eps,
                                   post_add=pa, want_f32=emit["f32"], want_g8=emit["g8"], out=dst)
            if timed:
                ev[1].record()
                self.tap_events.append(ev)
            res = list(res) if isinstance(res, tuple) else [res]
            out = {"f32": None, "g8": {}}
            if emit["f32"]:
                out["f32"] = res.pop(0)
            if emit["g8"]:
                out["g8"][ops.G8_F16] = res.pop(0)
            return out

        run.emit = emit
        run.parts = dict(segs=segs, terms=terms, act=act, leak=leak, pn=pn, post_add=post_add, prec=prec, cout=cout, out_hw=out_hw)
        run.info = {
            "kind": "conv2d_fused", "cout": cout, "act": act, "pixel_norm": pn, "prec": prec,
            "post_add": post_add.name if post_add is not None else None,
            "post_add_id": post_add.id if post_add is not None else None,
            "segments": [dict(src=src.name, src_id=src.id, c_off=c_off_src, cin=cin, up_log2=up, w_off=w_off,
                              kernel=tuple(term.conv.inputs[1].shape[:2]), weight=term.conv.inputs[1].attrs["var"])
                         for (src, c_off_src, up, term, w_off, cin) in segs],
        }
        return deps, run

    def plan_summary(self, fetch):
        """the launch plan of `fetch` as a list of dicts (no GPU needed): one entry per kernel launch"""
        plan = self._plans.get(fetch.id) or self._compile(fetch)
        self._plans[fetch.id] = plan
        out = []
        for node, fn in plan.steps:
            if node.op in ("variable", "placeholder"):
                continue
            info = dict(getattr(fn, "info", {"kind": node.op}))
            if info.get("kind") == "fused_into_next":
                continue
            info["node"] = node.name
            if hasattr(fn, "emit"):
                info["emit"] = dict(fn.emit)
            out.append(info)
        return out

    def _match_chain(self, n, single_use, top=True):
        """[pixel_norm]([act](lin)) -> (terms, act, leak, pn, eps)."""
        cur = n
        pn, pn_eps, act, leak = False, 1e-8, None, 0.2
        first = True

        def ok(node):
            return (first and top) or single_use(node)

        if cur.op == "pixel_norm" and ok(cur):
            pn, pn_eps = True, cur.attrs["eps"]
            cur = cur.inputs[0]
            first = False
        if cur.op == "act" and ok(cur):
            act, leak = cur.attrs["act"], cur.attrs.get("leak", 0.2)
            cur = cur.inputs[0]
            first = False
        if not ok(cur):
            return None
        terms = self._match_lin(cur, single_use, top=True)
        if terms is None:
            return None
        return terms, act, leak, pn, pn_eps

    def _segments_of(self, term):
        """Resolve the conv input into (source node, channel offset in source, up_log2, term,
        weight channel offset, channel count) tuples, looking through concat / nearest resize / slice."""
        out_h, out_w = term.conv.shape[1], term.conv.shape[2]
        out = []

        def resolve(node, c_lo, c_hi, w_off, up):
            # channels [c_lo, c_hi) of `node` feed weight channels starting at w_off
            if node.op == "concat":
                base = 0
                for part in node.inputs:
                    pc = part.shape[3]
                    lo, hi = max(c_lo, base), min(c_hi, base + pc)
                    if lo < hi:
                        if not resolve(part, lo - base, hi - base, w_off + (lo - c_lo), up):
                            return False
                    base += pc
                return True
            if node.op == "slice":
                b = node.attrs["begin"]
                return resolve(node.inputs[0], c_lo + b, c_hi + b, w_off, up)
            if node.op == "resize" and node.attrs["method"] == 1 and up == 0:
                src = node.inputs[0]
                fy, fx = node.attrs["oh"] // src.shape[1], node.attrs["ow"] // src.shape[2]
                if (fy == fx and _is_pow2(fy) and fy <= 16 and src.shape[1] * fy == node.attrs["oh"]
                        and src.shape[2] * fx == node.attrs["ow"]):
                    return resolve(src, c_lo, c_hi, w_off, fy.bit_length() - 1)
            if node.op == "reshape" and len(node.shape) == 4 and node.inputs[0].shape is not None \
                    and len(node.inputs[0].shape) == 4 and tuple(node.inputs[0].shape[1:]) == tuple(node.shape[1:]):
                return resolve(node.inputs[0], c_lo, c_hi, w_off, up)
            if len(node.shape) != 4 or (node.shape[1] << up) != out_h or (node.shape[2] << up) != out_w:
                return False
            out.append((node, c_lo, up, term, w_off, c_hi - c_lo))
            return True

        src = term.conv.inputs[0]
        if not resolve(src, 0, src.shape[3], 0, 0):
            return None
        return out

    # ---- parameter folding / packing ----------------------------------------
    def _bn_scale_shift(self, term):
        """(scale, shift) with y = scale * (conv + bias) + shift folded: returns per-channel
        scale s = gamma / sqrt(var + eps) and the effective bias s*(b - mean) + beta."""
        key = ("fold", term.conv.id)
        if key in self._folded:
            return self._folded[key]
        cout = term.conv.shape[-1]
        b = self.vars.get(term.bias.inputs[1].attrs["var"]).double() if term.bias is not None else \
            torch.zeros(cout, dtype=torch.float64, device=self.device)
        if term.bn is not None:
            gamma, beta, mean, var = [self.vars.get(v.attrs["var"]).double() for v in term.bn.inputs[1:5]]
            s = gamma / torch.sqrt(var + term.bn.attrs["eps"])
            eff = s * (b - mean) + beta
            res = (s.float().contiguous(), eff)
        else:
            res = (None, b)
        self._folded[key] = res
        return res

    def _packed_for(self, term, w_off, cin, prec):
        key = ("pack", term.conv.id, w_off, cin, prec)
        pk = self._packed.get(key)
        if pk is None:
            w = self.vars.get(term.conv.inputs[1].attrs["var"])
            scale, _ = self._bn_scale_shift(term)
            pk = ops.pack_conv_weights(w, wscale=term.conv.attrs["wscale"], cout_scale=scale, c_off=w_off, cin=cin,
                                       prec=prec)
            self._packed[key] = pk
        return pk

    def _bias_for(self, terms):
        key = ("bias",) + tuple(t.conv.id for t in terms)
        b = self._folded.get(key)
        if b is None:
            tot = None
            for t in terms:
                _, eff = self._bn_scale_shift(t)
                tot = eff if tot is None else tot + eff
            b = tot.float().contiguous()
            self._folded[key] = b
        return b

    # ---- one kernel per node ----------------------------------------------------
    def _fallback(self, n, single_use):
        op = n.op
        if op == "reshape":
            tgt = n.attrs["target"]
            return [n.inputs[0]], lambda env, i=n.inputs[0], t=tgt: self._f32(env, i).reshape(t)
        if op == "concat":
            return list(n.inputs), lambda env, ins=n.inputs: torch.cat([self._f32(env, i) for i in ins], dim=-1).contiguous()
        if op == "slice":
            b, s = n.attrs["begin"], n.attrs["size"]
            return [n.inputs[0]], lambda env, i=n.inputs[0]: self._f32(env, i)[..., b:b + s].contiguous()
        if op == "slice_flat":
            cnt = n.attrs["count"]
            return [n.inputs[0]], lambda env, i=n.inputs[0]: self._f32(env, i).reshape(self._f32(env, i).shape[0], -1)[:, :cnt].contiguous()
        if op == "add":
            a, b = n.inputs
            return [a, b], lambda env: ops.add_act(self._f32(env, a), self._f32(env, b))
        if op == "act":
            x = n.inputs[0]
            if x.op == "add" and single_use(x):
                a, b = x.inputs
                return [a, b], lambda env: ops.add_act(self._f32(env, a), self._f32(env, b), n.attrs["act"], n.attrs.get("leak", 0.2))
            direct = self._match_direct(n, single_use)
            if direct is not None:
                return direct
            return [x], lambda env: ops.add_act(self._f32(env, x), None, n.attrs["act"], n.attrs.get("leak", 0.2))
        if op == "pixel_norm":
            x = n.inputs[0]
            return [x], lambda env: ops.pixel_norm(self._f32(env, x), n.attrs["eps"])
        if op == "resize":
            x = n.inputs[0]
            return [x], lambda env: ops.resize_images(self._f32(env, x), n.attrs["oh"], n.attrs["ow"], n.attrs["method"])
        if op == "avg_pool":
            x = n.inputs[0]
            return [x], lambda env: ops.avg_pool2(self._f32(env, x))
        if op == "random_normal":
            x = n.inputs[0]

            def run_noise(env, n=n, x=x):
                ref = self._f32(env, x)
                gen = self.__dict__.setdefault("_noise_gen", {})
                if n.id not in gen:
                    seed = n.attrs["seed"] if n.attrs["seed"] is not None else n.id        # one stream per noise node
                    gen[n.id] = torch.Generator(device=ref.device).manual_seed(1000003 * seed + 17)
                shape = tuple(ref.shape[:-1]) + (n.shape[-1],)
                return torch.randn(shape, generator=gen[n.id], device=ref.device, dtype=torch.float32) * n.attrs["stddev"]

            return [x], run_noise
        if op == "max_pool":
            x = n.inputs[0]
            return [x], lambda env: ops.max_pool(self._f32(env, x), n.attrs["k"], n.attrs["s"])
        if op == "minibatch_stddev":
            x = n.inputs[0]
            return [x], lambda env: ops.minibatch_stddev(self._f32(env, x), n.attrs["group_size"])
        if op == "lerp":
            from . import train_ops
            t = n.attrs["t"]

            def run_lerp(env, n=n, t=t):
                tv = t.value(getattr(self, "_scalars", {})) if isinstance(t, G.Scalar) else float(t)
                xin = None if n.attrs["zero_x"] else self._f32(env, n.inputs[0])
                return train_ops.lerp(xin, self._f32(env, n.inputs[-1]), tv)

            return list(n.inputs), run_lerp
        if op == "depth_to_space":
            x = n.inputs[0]
            return [x], lambda env: ops.depth_to_space(self._f32(env, x), n.attrs["r"])
        if op == "advect":
            from . import train_ops

            def run_advect(env, n=n):
                src, vel = self._f32(env, n.inputs[0]), self._f32(env, n.inputs[1])
                flags = self._f32(env, n.inputs[2]) if len(n.inputs) > 2 else torch.zeros_like(src[..., :1])
                at = n.attrs
                return train_ops.advect(src, vel, flags, at["dt"], at["order"], at["strength"], at["start_bz"])

            return list(n.inputs), run_advect
        if op in ("conv2d", "conv2d_transpose", "bias_add", "batch_norm"):
            direct = self._match_direct(n, single_use)
            if direct is not None:
                return direct
        raise G.GraphError("no HIP lowering for node %r (inputs %r)" % (n, n.inputs))

    def _match_direct(self, n, single_use):
        """[act](bn?(bias_add?(conv2d | matmul))) on the vector-ALU kernel: strided convs, cout > 128, FC."""
        cur = n
        act, leak = None, 0.2
        if cur.op == "act":
            act, leak = cur.attrs["act"], cur.attrs.get("leak", 0.2)
            nxt = cur.inputs[0]
            if not single_use(nxt):
                return None
            cur = nxt
        bn = bias = None
        if cur.op == "batch_norm":
            if cur.attrs["training"]:
                raise NotImplementedError("training-mode batch norm is not lowered yet")
            bn = cur
            if not single_use(cur.inputs[0]):
                return None
            cur = cur.inputs[0]
        if cur.op == "bias_add":
            bias = cur
            if not single_use(cur.inputs[0]):
                return None
            cur = cur.inputs[0]
        if cur.op not in ("conv2d", "matmul", "conv2d_transpose"):
            return None
        conv = cur
        x, wv = conv.inputs
        term = _Term(conv, bias, bn)
        is_fc = conv.op == "matmul"

        def run(env):
            scale, eff = self._bn_scale_shift(term)
            b = eff.float().contiguous()
            w = self.vars.get(wv.attrs["var"])
            xin = self._f32(env, x)
            if conv.op == "conv2d_transpose":
                # GAN.deconvolutional_layer: batch norm folds into the filter's output-channel axis (axis 2 of [kh,kw,cout,cin])
                wf = w if scale is None else (w * scale.view(1, 1, -1, 1)).contiguous()
                return ops.conv2d_transpose(xin, wf, conv.attrs["stride"], conv.attrs["wscale"], b, act, leak, prec=self.prec)
            if is_fc and scale is None:
                from . import train_ops
                return train_ops.fc_forward(xin.contiguous(), w, conv.attrs["wscale"], b, act, leak)
            if is_fc:
                y = ops.conv2d_direct(xin.reshape(xin.shape[0], 1, 1, xin.shape[1]).contiguous(),
                                      w.reshape(1, 1, w.shape[0], w.shape[1]), (1, 1), conv.attrs["wscale"], scale, b,
                                      act, leak)
                return y.reshape(xin.shape[0], w.shape[1])
            return ops.conv2d_direct(xin, w, conv.attrs["stride"], conv.attrs["wscale"], scale, b, act, leak)

        return [x], run
