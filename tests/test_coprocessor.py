"""Event-driven operator boundary (fhe_amd/coprocessor.py): dependency waves, batching of like
events, scalar byte, handle types, casts, rand.  CPU: on the plaintext mock backend, replaying the
reference's operator vectors as executor events; GPU: a block of ERC20-style transfers."""
import json
import os

import pytest

from fhe_amd.coprocessor import BINARY, Coprocessor, Event, TYPE_WIDTH, handle_type
from oracle import radix
from tests.mock_engine import FusingMockBackend, MockBackend

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(GOLDEN, "fhevm_operator_kats.json")))["kats"]
WIDTH_TYPE = {w: t for t, w in TYPE_WIDTH.items()}
OP_EVENT = {}
for _ev, _op in BINARY.items():
    OP_EVENT.setdefault(_op, _ev)
OP_EVENT.update({"neg": "FheNeg", "not": "FheNot"})


def mk_handle(counter: int, width: int) -> bytes:
    h = bytearray(32)
    h[0:8] = counter.to_bytes(8, "big")
    h[30] = WIDTH_TYPE[width]
    return bytes(h)


def kat_events(backend, cop, kats):
    """Each vector as the events the fhevm library emits for it: operands promoted to the wider
    type by Cast, clear right operands through scalarByte = 1, clear left operands trivially
    encrypted (the library only has scalar-right overloads)."""
    events, expect = [], []
    n = 0
    for k in kats:
        op, w, ins, exp = radix.kat_case(k)
        enc = [t.startswith("e") for t in k["types"]]
        if op in ("shl", "shr", "rotl", "rotr", "div", "rem"):
            w = radix.type_width(k["types"][0])
        hs = []
        for i, v in enumerate(ins):
            wi = radix.type_width(k["types"][i])
            if enc[i]:
                n += 1
                h = mk_handle(n, wi)
                cop.put(h, backend.encrypt(v, wi))
                tw = 8 if (i == 1 and op in ("shl", "shr", "rotl", "rotr")) else w
                if wi != tw:
                    n += 1
                    h2 = mk_handle(n, tw)
                    events.append(Event("Cast", {"ct": h, "toType": WIDTH_TYPE[tw], "result": h2}))
                    h = h2
                hs.append(h)
            elif i == 0:
                n += 1
                h = mk_handle(n, w)
                events.append(Event("TrivialEncrypt", {"pt": v, "toType": WIDTH_TYPE[w], "result": h}))
                hs.append(h)
            else:
                hs.append(v)
        n += 1
        out_w = 1 if op in ("eq", "ne", "ge", "gt", "le", "lt") else w
        res = mk_handle(n, out_w)
        if len(ins) == 1:
            events.append(Event(OP_EVENT[op], {"ct": hs[0], "result": res}))
        else:
            scalar = not enc[1]
            rhs = hs[1].to_bytes(32, "big") if scalar else hs[1]
            events.append(Event(OP_EVENT[op], {"lhs": hs[0], "rhs": rhs, "scalarByte": b"\x01" if scalar else b"\x00",
                                               "result": res}))
        expect.append((res, exp))
    return events, expect


def test_every_reference_vector_as_executor_events():
    backend = MockBackend()
    cop = Coprocessor(backend)
    events, expect = kat_events(backend, cop, KATS)
    waves = cop.execute(events)
    assert waves <= 3                                  # trivial encryptions / casts, then the operators
    for h, exp in expect:
        assert backend.decrypt(cop.get(h)) == exp
        assert handle_type(h) in TYPE_WIDTH
    # like events were batched: far fewer engine calls than events
    assert cop.executed == len(events) and backend.batches < 400 < len(KATS)


def test_dependency_waves_select_cast_and_rand():
    backend = MockBackend()
    cop = Coprocessor(backend)
    H = lambda i, w: mk_handle(i, w)
    cop.put(H(1, 64), backend.encrypt(10000, 64))      # balance
    cop.put(H(2, 64), backend.encrypt(1337, 64))       # amount
    cop.put(H(3, 64), backend.encrypt(5, 64))          # recipient balance
    ev = [
        # deliberately out of order: the scheduler sorts by availability of the inputs
        Event("FheSub", {"lhs": H(1, 64), "rhs": H(5, 64), "scalarByte": b"\x00", "result": H(6, 64)}),
        Event("FheIfThenElse", {"control": H(4, 1), "ifTrue": H(2, 64), "ifFalse": H(9, 64), "result": H(5, 64)}),
        Event("FheLe", {"lhs": H(2, 64), "rhs": H(1, 64), "scalarByte": b"\x00", "result": H(4, 1)}),
        Event("TrivialEncrypt", {"pt": 0, "toType": 5, "result": H(9, 64)}),
        Event("FheAdd", {"lhs": H(3, 64), "rhs": H(5, 64), "scalarByte": b"\x00", "result": H(7, 64)}),
        Event("Cast", {"ct": H(7, 64), "toType": 4, "result": H(8, 32)}),
        Event("Cast", {"ct": H(7, 64), "toType": 0, "result": H(10, 1)}),
        Event("FheRandBounded", {"upperBound": 1 << 10, "randType": 3, "seed": bytes(16), "result": H(11, 16)}),
        Event("FheRand", {"randType": 2, "seed": bytes(range(16)), "result": H(12, 8)}),
    ]
    waves = cop.execute(ev)
    assert waves == 4                                  # le|trivial|rand -> select -> sub|add -> casts
    d = lambda h: backend.decrypt(cop.get(h))
    assert (d(H(6, 64)), d(H(7, 64)), d(H(8, 32)), d(H(10, 1))) == (10000 - 1337, 1342, 1342, 1)
    assert d(H(11, 16)) < 1 << 10
    with pytest.raises(KeyError):
        cop.execute([Event("FheNeg", {"ct": H(99, 8), "result": H(100, 8)})])
    with pytest.raises(ValueError):
        cop.execute([Event("FheRandBounded", {"upperBound": 1000, "randType": 3, "seed": bytes(16), "result": H(13, 16)})])


def _transfer_block(cop, enc, n, new):
    """n transfers as events (le, select against a trivially encrypted zero, sub, add)."""
    ev, outs, plain = [], [], []
    zero = new(64)
    ev.append(Event("TrivialEncrypt", {"pt": 0, "toType": 5, "result": zero}))
    for i in range(n):
        bal, to, amt = 1000 + 37 * i, 5 * i, (400 * i) % 1500
        hb, ht, ha = new(64), new(64), new(64)
        cop.put(hb, enc(bal, 64)); cop.put(ht, enc(to, 64)); cop.put(ha, enc(amt, 64))
        ok, v, nb, nt = new(1), new(64), new(64), new(64)
        ev += [Event("FheLe", {"lhs": ha, "rhs": hb, "scalarByte": b"\x00", "result": ok}),
               Event("FheIfThenElse", {"control": ok, "ifTrue": ha, "ifFalse": zero, "result": v}),
               Event("FheSub", {"lhs": hb, "rhs": v, "scalarByte": b"\x00", "result": nb}),
               Event("FheAdd", {"lhs": ht, "rhs": v, "scalarByte": b"\x00", "result": nt})]
        moved = amt if amt <= bal else 0
        plain.append((bal - moved, to + moved, int(amt <= bal), moved))
        outs.append((nb, nt, ok, v))
    return ev, outs, plain


def test_transactions_of_one_shape_are_fused():
    backend = FusingMockBackend()
    cop = Coprocessor(backend)
    c = [0]

    def new(w):
        c[0] += 1
        return mk_handle(c[0], w)
    ev, outs, plain = _transfer_block(cop, backend.encrypt, 9, new)
    # plus one transaction of another shape and a lone event
    x, y, r1, r2, r3 = new(8), new(8), new(8), new(8), new(8)
    cop.put(x, backend.encrypt(200, 8)); cop.put(y, backend.encrypt(100, 8))
    ev += [Event("FheAdd", {"lhs": x, "rhs": y, "scalarByte": b"\x00", "result": r1}),
           Event("FheMul", {"lhs": r1, "rhs": (3).to_bytes(32, "big"), "scalarByte": b"\x01", "result": r2}),
           Event("FheNot", {"ct": y, "result": r3})]
    cop.execute(ev)
    d = lambda h: backend.decrypt(cop.get(h))
    assert [tuple(d(h) for h in o) for o in outs] == plain          # intermediates (ok, v) are kept too
    assert (d(r1), d(r2), d(r3)) == (44, 132, 155)
    assert backend.fused_calls == 2                                 # nine transfers in one call, the 8-bit chain in another
    assert cop.executed == len(ev)
    # with fusion off the same block goes wave by wave
    b2 = FusingMockBackend()
    cop2 = Coprocessor(b2, fuse=False)
    c[0] = 0
    ev2, outs2, plain2 = _transfer_block(cop2, b2.encrypt, 9, new)
    assert cop2.execute(ev2) == 3 and getattr(b2, "fused_calls", 0) == 0
    assert [tuple(b2.decrypt(cop2.get(h)) for h in o) for o in outs2] == plain2


def test_resident_store_moves_only_inputs_and_requested_results():
    """With a backend that keeps ciphertexts under handles the worker runs the events on handles: the
    same results, and nothing crosses the bus between events."""
    from tests.mock_engine import ResidentMockBackend
    backend = ResidentMockBackend()
    cop = Coprocessor(backend)
    assert cop.resident
    c = [0]

    def new(w):
        c[0] += 1
        return mk_handle(c[0], w)
    ev, outs, plain = _transfer_block(cop, backend.encrypt, 9, new)
    x, y, r1, r2, r3, r4 = new(8), new(8), new(8), new(8), new(8), new(16)
    cop.put(x, backend.encrypt(200, 8)); cop.put(y, backend.encrypt(100, 8))
    ev += [Event("FheAdd", {"lhs": x, "rhs": y, "scalarByte": b"\x00", "result": r1}),
           Event("FheMul", {"lhs": r1, "rhs": (3).to_bytes(32, "big"), "scalarByte": b"\x01", "result": r2}),
           Event("FheNot", {"ct": y, "result": r3}),
           Event("Cast", {"ct": r2, "toType": 3, "result": r4})]
    uploads = backend.uploads
    cop.execute(ev)
    # the cast is made on the host (one download, one upload); everything else stayed where it was
    assert backend.uploads - uploads <= 2 and backend.downloads <= 1
    d = lambda h: backend.decrypt(cop.get(h))
    assert [tuple(d(h) for h in o) for o in outs] == plain
    assert (d(r1), d(r2), d(r3), d(r4)) == (44, 132, 155, 132)
    assert backend.fused_calls == 2 and cop.executed == len(ev)
    with pytest.raises(KeyError):
        cop.get(new(8))
    # the reference vectors through the handle route, too
    b2 = ResidentMockBackend()
    cop2 = Coprocessor(b2)
    events, expect = kat_events(b2, cop2, KATS[:600])
    cop2.execute(events)
    for h, exp in expect:
        assert b2.decrypt(cop2.get(h)) == exp


@pytest.mark.gpu
def test_events_on_handles_equal_the_host_buffer_route_bit_for_bit():
    """A chain of events over handles (ciphertexts resident in device memory) against the same chain
    with every operand and result crossing PCIe: identical ciphertext bytes, identical store traffic
    (tests/fhevm-suite/e2e/test/utils.ts:182-214: handles in, handle out)."""
    from fhe_amd.native import Context, LfheError
    ctx = Context(seed=bytes(range(32)), params_id=3)
    ctx.keygen()
    try:
        ctx.reseed_encryption(bytes(range(32)))
        c = [0]

        def new(w):
            c[0] += 1
            return mk_handle(c[0], w)
        cops = [Coprocessor(ctx, resident=False), Coprocessor(ctx, resident=True)]
        assert cops[1].resident and not cops[0].resident
        n = 5
        ev, outs, inputs = [], [], []
        zero = new(64)
        ev.append(Event("TrivialEncrypt", {"pt": 0, "toType": 5, "result": zero}))
        for i in range(n):
            bal, to, amt = 1000 + 37 * i, 5 * i, (700 * i) % 1500
            hb, ht, ha = new(64), new(64), new(64)
            inputs += [(hb, ctx.encrypt(bal, 64)), (ht, ctx.encrypt(to, 64)), (ha, ctx.encrypt(amt, 64))]
            ok, v, nb, nt = new(1), new(64), new(64), new(64)
            ev += [Event("FheLe", {"lhs": ha, "rhs": hb, "scalarByte": b"\x00", "result": ok}),
                   Event("FheIfThenElse", {"control": ok, "ifTrue": ha, "ifFalse": zero, "result": v}),
                   Event("FheSub", {"lhs": hb, "rhs": v, "scalarByte": b"\x00", "result": nb}),
                   Event("FheAdd", {"lhs": ht, "rhs": v, "scalarByte": b"\x00", "result": nt})]
            outs += [ok, v, nb, nt]
        # a second wave on the results: lone events (not fused), clear operands, a shift by a clear amount, a unary
        x, y = outs[2], outs[6]
        m1, m2, m3, m4, m5 = new(64), new(1), new(64), new(64), new(64)
        ev += [Event("FheMul", {"lhs": x, "rhs": (1000003).to_bytes(32, "big"), "scalarByte": b"\x01", "result": m1}),
               Event("FheGt", {"lhs": x, "rhs": y, "scalarByte": b"\x00", "result": m2})]
        ev2 = [Event("FheShl", {"lhs": m1, "rhs": (13).to_bytes(32, "big"), "scalarByte": b"\x01", "result": m3}),
               Event("FheNeg", {"ct": y, "result": m4}),
               Event("FheBitXor", {"lhs": m1, "rhs": y, "scalarByte": b"\x00", "result": m5})]
        outs += [m1, m2, m3, m4, m5]
        for cop in cops:
            for h, ct in inputs:
                cop.put(h, ct)
            cop.execute(ev)
            cop.execute(ev2)
        for h in outs:
            assert cops[0].get(h) == cops[1].get(h), h.hex()
        bal0, amt1 = 1000, 700
        assert ctx.decrypt(cops[1].get(outs[2])) == bal0 and ctx.decrypt(cops[1].get(outs[6])) == 1037 - amt1
        assert ctx.decrypt(cops[1].get(m3)) == ((bal0 * 1000003) << 13) & ((1 << 64) - 1)
        # the store itself: sizes, replacement, errors
        entries, held = ctx.store_stat()
        assert entries == len(inputs) + 1 + len(outs) and held >= entries * 1 * 16392
        ctx.store_put(m4, ctx.store_get(m5))            # replace an entry
        assert ctx.store_get(m4) == cops[0].get(m5)
        ctx.store_drop(m4)
        assert not ctx.store_has(m4) and ctx.store_stat()[0] == entries - 1
        with pytest.raises(LfheError) as e:
            ctx.evaluate_batch_handles("add", 64, [m4], [m5], None, [new(64)])    # unknown handle
        assert e.value.code == -1
        with pytest.raises(LfheError) as e:
            ctx.evaluate_batch_handles("add", 32, [m5], [m5], None, [new(32)])    # stored width differs
        assert e.value.code == -1
        assert ctx.store_stat()[0] == entries - 1          # a failed call stores nothing
    finally:
        ctx.close()


@pytest.mark.gpu
def test_block_of_transfers_and_rand_on_gpu():
    """A block of transfers (le, select, sub, add per transfer: EncryptedERC20.sol:87-93,208-220)
    as executor events: four waves, one batched engine call per (wave, operation)."""
    from fhe_amd.native import Context
    ctx = Context(seed=bytes(range(32)), params_id=2)
    ctx.keygen()
    try:
        cop = Coprocessor(ctx)
        n, c = 6, 0
        plain, ev, outs = [], [], []

        def new(w):
            nonlocal c
            c += 1
            return mk_handle(c, w)
        zero = new(64)
        ev.append(Event("TrivialEncrypt", {"pt": 0, "toType": 5, "result": zero}))
        for i in range(n):
            bal, to, amt = 1000 + 37 * i, 5 * i, (400 * i) % 1500
            hb, ht, ha = new(64), new(64), new(64)
            cop.put(hb, ctx.encrypt(bal, 64)); cop.put(ht, ctx.encrypt(to, 64)); cop.put(ha, ctx.encrypt(amt, 64))
            ok, v, nb, nt = new(1), new(64), new(64), new(64)
            ev += [Event("FheLe", {"lhs": ha, "rhs": hb, "scalarByte": b"\x00", "result": ok}),
                   Event("FheIfThenElse", {"control": ok, "ifTrue": ha, "ifFalse": zero, "result": v}),
                   Event("FheSub", {"lhs": hb, "rhs": v, "scalarByte": b"\x00", "result": nb}),
                   Event("FheAdd", {"lhs": ht, "rhs": v, "scalarByte": b"\x00", "result": nt})]
            moved = amt if amt <= bal else 0
            plain.append((bal - moved, to + moved))
            outs.append((nb, nt))
        before = cop.launches
        cop.execute(ev)
        assert cop.launches - before == 2              # the trivial zero, then the six transfers as one fused schedule
        assert [(ctx.decrypt(cop.get(a)), ctx.decrypt(cop.get(b))) for a, b in outs] == plain
        # oblivious randoms: deterministic in the seed, inside the bound, not constant
        r1 = ctx.decrypt(ctx.rand(64, bytes(range(16))))
        assert r1 == ctx.decrypt(ctx.rand(64, bytes(range(16))))
        vals = [ctx.decrypt(ctx.rand(16, bytes([i]) * 16, 11)) for i in range(24)]
        assert all(v < 1 << 11 for v in vals) and len(set(vals)) > 12
        ones = sum(bin(v).count("1") for v in vals)
        assert 24 * 11 * 0.3 < ones < 24 * 11 * 0.7
        assert ctx.decrypt(ctx.rand(1, bytes(16))) in (0, 1)
    finally:
        ctx.close()
