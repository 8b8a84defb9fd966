"""Event-driven operator boundary: the fhevm executor emits one event per operation with bytes32
handles (tests/fhevm-suite/e2e/test/utils.ts:182-214); a coprocessor worker resolves the handles to
ciphertexts, runs the operations and stores the results under the result handles.

What this adds over calling the engine once per event: the events of a block are sorted into waves
by data dependence, and within a wave all events of one (operation, type, scalar-ness) become ONE
`evaluate_batch` call, so every bootstrap layer of that operation is one GPU launch for the lot.

The type of a ciphertext is byte 30 of its handle (utils.ts:283-285, FhevmHandle.ts:213).
VerifyCiphertext (input proofs) is outside this backend.
"""
from __future__ import annotations

from dataclasses import dataclass, field

# FheType ids (sdk/relayer/src/sdk/FheType.ts:17-27; tests/fhevm-suite/e2e/test/types.ts)
TYPE_WIDTH = {0: 1, 2: 8, 3: 16, 4: 32, 5: 64, 6: 128, 7: 160, 8: 256}

BINARY = {"FheAdd": "add", "FheSub": "sub", "FheMul": "mul", "FheDiv": "div", "FheRem": "rem",
          "FheBitAnd": "and", "FheBitOr": "or", "FheBitXor": "xor", "FheShl": "shl", "FheShr": "shr",
          "FheRotl": "rotl", "FheRotr": "rotr", "FheEq": "eq", "FheNe": "ne", "FheGe": "ge", "FheGt": "gt",
          "FheLe": "le", "FheLt": "lt", "FheMin": "min", "FheMax": "max",
          "FheEqBytes": "eq", "FheNeBytes": "ne"}
UNARY = {"FheNeg": "neg", "FheNot": "not"}


def handle_type(handle: bytes) -> int:
    if len(handle) != 32:
        raise ValueError("a handle is 32 bytes")
    return handle[30]


def handle_width(handle: bytes) -> int:
    t = handle_type(handle)
    if t not in TYPE_WIDTH:
        raise ValueError(f"unsupported FheType {t}")
    return TYPE_WIDTH[t]


@dataclass
class Event:
    """One executor event: `name` as in the ABI, `args` by the ABI's argument names
    (lhs, rhs, scalarByte, result, ct, toType, pt, control, ifTrue, ifFalse, randType, seed, upperBound)."""
    name: str
    args: dict = field(default_factory=dict)

    def inputs(self) -> list:
        a = self.args
        if self.name in BINARY:
            return [a["lhs"]] if _is_scalar(a) else [a["lhs"], a["rhs"]]
        if self.name in UNARY or self.name == "Cast":
            return [a["ct"]]
        if self.name == "FheIfThenElse":
            return [a["control"], a["ifTrue"], a["ifFalse"]]
        return []


def _is_scalar(a: dict) -> bool:
    sb = a.get("scalarByte", b"\x00")
    return (sb if isinstance(sb, int) else sb[0]) == 1


def _clear(v) -> int:
    return v if isinstance(v, int) else int.from_bytes(v, "big")


class _DeviceStore:
    """The backend's own store of ciphertexts in device memory, with the mapping interface the worker
    uses: writing uploads, reading downloads, operator results are entered by the backend itself
    (`mark`).  Ciphertexts cross PCIe when they enter (inputs) and when somebody asks for them
    (decryption requests), not between events."""

    def __init__(self, backend):
        self.b = backend
        self.known: set = set()

    def __contains__(self, h) -> bool:
        return bytes(h) in self.known

    def __getitem__(self, h) -> bytes:
        if bytes(h) not in self.known:
            raise KeyError(bytes(h).hex())
        return self.b.store_get(bytes(h))

    def __setitem__(self, h, ct: bytes):
        self.b.store_put(bytes(h), ct)
        self.known.add(bytes(h))

    def update(self, d: dict):
        for k, v in d.items():
            self[k] = v

    def mark(self, h):
        self.known.add(bytes(h))

    def __len__(self) -> int:
        return len(self.known)


class Coprocessor:
    """backend: an object with evaluate_batch(op, width, lhs[], rhs[]|None, cond[]|None),
    evaluate(op, width, lhs, rhs, cond), encrypt(value, width, trivial=True) and
    rand(width, seed16, random_bits) -- fhe_amd.native.Context is one.
    resident: keep the ciphertexts in the backend's device store and run the events on handles
    (store_put / store_get / evaluate_batch_handles / evaluate_fused_handles); the default when the
    backend has such a store.  Results are bit-identical to the host-buffer route."""

    def __init__(self, backend, fuse: bool = True, resident: bool | None = None):
        self.backend = backend
        self.resident = hasattr(backend, "evaluate_batch_handles") if resident is None else resident
        self.store = _DeviceStore(backend) if self.resident else {}
        self.launches = 0          # evaluate_batch / evaluate_fused / single calls issued
        self.executed = 0
        self.fuse = fuse and hasattr(backend, "evaluate_fused")

    def put(self, handle: bytes, ciphertext: bytes):
        self.store[bytes(handle)] = ciphertext

    def get(self, handle: bytes) -> bytes:
        return self.store[bytes(handle)]

    # ---- fusion of transactions of one shape
    def _fuse(self, events: list) -> list:
        """Operator events that form chains (a transaction's le -> select -> sub / add) are grouped
        into connected components over the handles produced inside the list; components of the same
        shape -- same operators, widths, clear operands and wiring -- are run as ONE fused schedule
        over all of them, every result handle kept.  Returns the events that were not fused."""
        ops = [e for e in events if e.name in BINARY or e.name in UNARY or e.name == "FheIfThenElse"]
        producer = {bytes(e.args["result"]): i for i, e in enumerate(ops)}
        parent = list(range(len(ops)))

        def find(i):
            while parent[i] != i:
                parent[i] = parent[parent[i]]
                i = parent[i]
            return i
        for i, e in enumerate(ops):
            for h in e.inputs():
                j = producer.get(bytes(h))
                if j is not None:
                    parent[find(i)] = find(j)
        comps: dict = {}
        for i in range(len(ops)):
            comps.setdefault(find(i), []).append(i)
        shapes: dict = {}
        for members in comps.values():
            if len(members) < 2:
                continue                      # a lone event gains nothing over the batched path
            inputs, ref, steps, ok = [], {}, [], True
            for i in members:                 # list order is a topological order of a valid block
                e, a = ops[i], ops[i].args
                for h in e.inputs():
                    hb = bytes(h)
                    if hb not in ref:
                        if hb in producer or hb not in self.store:
                            ok = False        # produced later in the list, or by an event outside the operators
                            break
                        ref[hb] = len(inputs)
                        inputs.append(hb)
                if not ok:
                    break
                if e.name in BINARY:
                    rhs = ("clear", _clear(a["rhs"])) if _is_scalar(a) else ref[bytes(a["rhs"])]
                    steps.append((BINARY[e.name], handle_width(a["lhs"]), ref[bytes(a["lhs"])], rhs, None))
                elif e.name in UNARY:
                    steps.append((UNARY[e.name], handle_width(a["ct"]), ref[bytes(a["ct"])], None, None))
                else:
                    steps.append(("select", handle_width(a["ifTrue"]), ref[bytes(a["ifTrue"])], ref[bytes(a["ifFalse"])],
                                  ref[bytes(a["control"])]))
                ref[bytes(a["result"])] = -1 - len(steps)      # placeholder: step results are renumbered below
            if not ok:
                continue
            n_in = len(inputs)
            fix = lambda r: r if not isinstance(r, int) or r >= 0 else n_in + (-1 - r) - 1
            steps = tuple((op, w, fix(l), r if isinstance(r, tuple) or r is None else fix(r), None if c is None else fix(c))
                          for op, w, l, r, c in steps)
            widths = tuple(handle_width(h) for h in inputs)
            shapes.setdefault((steps, widths), []).append((members, inputs))
        fused_ids = set()
        for (steps, widths), insts in shapes.items():
            outs = list(range(len(widths), len(widths) + len(steps)))
            if self.resident:
                cols = [[inp[k] for _, inp in insts] for k in range(len(widths))]
                out_h = [[bytes(ops[members[sidx]].args["result"]) for members, _ in insts] for sidx in range(len(steps))]
                self.backend.evaluate_fused_handles(list(steps), list(widths), cols, outs, out_h)
                self.launches += 1
                for members, _ in insts:
                    for i in members:
                        self.store.mark(ops[i].args["result"])
                        fused_ids.add(id(ops[i]))
                self.executed += sum(len(m) for m, _ in insts)
                continue
            cols = [[self.store[inp[k]] for _, inp in insts] for k in range(len(widths))]
            out_w = [1 if op in ("eq", "ne", "ge", "gt", "le", "lt") else w for op, w, _, _, _ in steps]
            res = self.backend.evaluate_fused(list(steps), list(widths), cols, outs, out_w)
            self.launches += 1
            for k, (members, _) in enumerate(insts):
                for sidx, i in enumerate(members):
                    self.store[bytes(ops[i].args["result"])] = res[sidx][k]
                    fused_ids.add(id(ops[i]))
            self.executed += sum(len(m) for m, _ in insts)
        return [e for e in events if id(e) not in fused_ids]

    # ---- scheduling
    def execute(self, events: list) -> int:
        """Runs all events; returns the number of waves.  Raises KeyError if an input handle is
        neither stored nor produced by another event of the list."""
        pending = list(events)
        waves = 0
        if self.fuse:
            # first the events that only make inputs (trivial encryptions, casts, randoms), then the
            # operator chains as fused schedules, then whatever is left wave by wave
            makers = [e for e in pending if e.name not in BINARY and e.name not in UNARY and e.name != "FheIfThenElse"
                      and all(bytes(h) in self.store for h in e.inputs())]
            for e in makers:
                self.store.update(self._run_group(self._group_key(e), [e]))
            self.executed += len(makers)
            done = set(map(id, makers))
            pending = self._fuse([e for e in pending if id(e) not in done])
            waves = 1 if len(pending) < len(events) else 0
        while pending:
            ready = [e for e in pending if all(bytes(h) in self.store for h in e.inputs())]
            if not ready:
                missing = next(h for e in pending for h in e.inputs() if bytes(h) not in self.store)
                raise KeyError(f"handle {bytes(missing).hex()} is not available")
            # results of this wave become visible only after the wave: a later event of the same
            # wave never reads them (it would not be ready)
            results = {}
            groups: dict = {}
            for e in ready:
                groups.setdefault(self._group_key(e), []).append(e)
            for key, evs in groups.items():
                results.update(self._run_group(key, evs))
            self.store.update(results)
            done = set(map(id, ready))
            pending = [e for e in pending if id(e) not in done]
            self.executed += len(ready)
            waves += 1
        return waves

    def _group_key(self, e: Event):
        a = e.args
        if e.name in BINARY:
            return ("bin", BINARY[e.name], handle_width(a["lhs"]), _is_scalar(a))
        if e.name in UNARY:
            return ("un", UNARY[e.name], handle_width(a["ct"]))
        if e.name == "FheIfThenElse":
            return ("sel", handle_width(a["ifTrue"]))
        return (e.name, id(e))     # executed one by one

    def _run_group_resident(self, key, evs) -> dict:
        """The same on handles: operands and results stay in the backend's device store."""
        B = self.backend
        kind = key[0]
        res_h = [bytes(e.args["result"]) for e in evs]
        if kind == "bin":
            _, op, width, scalar = key
            if not scalar and op in ("shl", "shr", "rotl", "rotr"):
                for e in evs:
                    if handle_width(e.args["rhs"]) != 8:
                        raise ValueError("shift and rotation amounts are euint8")
            rhs = [_clear(e.args["rhs"]) if scalar else bytes(e.args["rhs"]) for e in evs]
            B.evaluate_batch_handles(op, width, [bytes(e.args["lhs"]) for e in evs], rhs, None, res_h)
        elif kind == "un":
            _, op, width = key
            B.evaluate_batch_handles(op, width, [bytes(e.args["ct"]) for e in evs], None, None, res_h)
        elif kind == "sel":
            B.evaluate_batch_handles("select", key[1], [bytes(e.args["ifTrue"]) for e in evs],
                                     [bytes(e.args["ifFalse"]) for e in evs], [bytes(e.args["control"]) for e in evs], res_h)
        else:
            (e,) = evs
            self.launches += 1
            return {res_h[0]: self._run_single(e)}     # made on the host (trivial encryption, cast, random): uploaded by the caller
        self.launches += 1
        for h in res_h:
            self.store.mark(h)
        return {}

    def _run_group(self, key, evs) -> dict:
        if self.resident:
            return self._run_group_resident(key, evs)
        B = self.backend
        out = {}
        kind = key[0]
        if kind == "bin":
            _, op, width, scalar = key
            lhs = [self.get(e.args["lhs"]) for e in evs]
            rhs = [_clear(e.args["rhs"]) if scalar else self.get(e.args["rhs"]) for e in evs]
            if not scalar and op in ("shl", "shr", "rotl", "rotr"):
                for e in evs:
                    if handle_width(e.args["rhs"]) != 8:
                        raise ValueError("shift and rotation amounts are euint8")
            res = B.evaluate_batch(op, width, lhs, rhs, None)
            self.launches += 1
            for e, r in zip(evs, res):
                out[bytes(e.args["result"])] = r
        elif kind == "un":
            _, op, width = key
            res = B.evaluate_batch(op, width, [self.get(e.args["ct"]) for e in evs], None, None)
            self.launches += 1
            for e, r in zip(evs, res):
                out[bytes(e.args["result"])] = r
        elif kind == "sel":
            width = key[1]
            res = B.evaluate_batch("select", width, [self.get(e.args["ifTrue"]) for e in evs],
                                   [self.get(e.args["ifFalse"]) for e in evs],
                                   [self.get(e.args["control"]) for e in evs])
            self.launches += 1
            for e, r in zip(evs, res):
                out[bytes(e.args["result"])] = r
        else:
            (e,) = evs
            out[bytes(e.args["result"])] = self._run_single(e)
            self.launches += 1
        return out

    def _run_single(self, e: Event) -> bytes:
        B, a = self.backend, e.args
        if e.name in ("TrivialEncrypt", "TrivialEncryptBytes"):
            return B.encrypt(_clear(a["pt"]), TYPE_WIDTH[a["toType"]], trivial=True)
        if e.name == "Cast":
            to_w = TYPE_WIDTH[a["toType"]]
            src = self.get(a["ct"])
            if to_w == 1 and handle_width(a["ct"]) != 1:        # integer -> ebool is "!= 0"
                return B.evaluate("ne", handle_width(a["ct"]), src, 0, None)
            return B.evaluate("cast", to_w, src, None, None)
        if e.name == "FheRand":
            w = TYPE_WIDTH[a["randType"]]
            return B.rand(w, bytes(a["seed"]), w)
        if e.name == "FheRandBounded":
            w = TYPE_WIDTH[a["randType"]]
            bound = _clear(a["upperBound"])
            if bound < 2 or bound & (bound - 1) or bound.bit_length() - 1 > w:
                raise ValueError("upperBound must be a power of two within the type")   # as contracts/Rand.sol requires
            return B.rand(w, bytes(a["seed"]), bound.bit_length() - 1)
        raise ValueError(f"event {e.name} is not handled by this backend")
