'use strict';
// The reference's plumbing case through the in-process binding (e2e/test/fhe.test.ts:88-175,
// local-dev/demo.ts:44-81): 42 and 17 as euint8 -> add / sub / lt / eq -> decrypt.
// Prints one JSON line; exits non-zero on any mismatch.  Needs a GPU.
const { LuxFHEClient, NativeEngine } = require('./index.js');

(async () => {
  const engine = new NativeEngine({ seed: Buffer.from(Array.from({ length: 32 }, (_, i) => (5 * i + 1) & 0xff)) });
  const sdk = new LuxFHEClient({ engine });
  await sdk.initialize();
  const a = await sdk.encrypt_uint8(42);
  const b = await sdk.encrypt_uint8(17);
  const res = {
    health: engine.health().status,
    publicKeyBytes: (await sdk.getPublicKey()).length,
    add: String(await sdk.decrypt(await sdk.evaluate('add', a, b, 8))),
    sub: String(await sdk.decrypt(await sdk.evaluate('sub', a, b, 8))),
    lt: String(await sdk.decrypt(await sdk.evaluate('lt', a, b, 8))),
    eq: String(await sdk.decrypt(await sdk.evaluate('eq', a, await sdk.encrypt_uint8(42), 8))),
    addScalar: String(await sdk.decrypt(await sdk.evaluate('add', a, 200, 8))),
    shl: String(await sdk.decrypt(await sdk.evaluate('shl', a, 3, 8))),
    mul64: String(await sdk.decrypt(await sdk.evaluate('mul', await sdk.encrypt_uint64(4293652672n), await sdk.encrypt_uint64(4293367238n), 64))),
    batch: engine.evaluateBatch('gt', 8, [a, b], [b, a]).map((c) => String(engine.decrypt(c))),
    rand: String(engine.decrypt(engine.rand(16, Buffer.alloc(16, 7), 10))),
    pbs: engine.pbsCount(),
  };
  // sixteen concurrent evaluate promises (the reference's tests fan out with Promise.all,
  // tests/fhevm-suite/e2e/test/instance.ts:32-36): they run off the event loop and share their
  // bootstrap launches, so the lot costs little more than one
  const conc = async (op, bits, n, vx, vy, model) => {
    const xs = [], ys = [];
    for (let i = 0; i < n; i++) { xs.push(await sdk.encrypt(vx(i), bits)); ys.push(await sdk.encrypt(vy(i), bits)); }
    await sdk.evaluate(op, xs[0], ys[0], bits);                       // schedule built and uploaded
    let best = Infinity;
    for (let k = 0; k < 3; k++) {
      const t = process.hrtime.bigint();
      await sdk.evaluate(op, xs[0], ys[0], bits);
      best = Math.min(best, Number(process.hrtime.bigint() - t) / 1e6);
    }
    // a timing statement about a shared machine: the best of up to four rounds, every round's values checked
    let all = Infinity, ticks = 0, ok = true;
    for (let round = 0; round < 4 && !(all < 1.8 * best); round++) {
      let turned = 0;
      const timer = setInterval(() => { turned++; }, 2);              // the event loop keeps turning meanwhile
      const t0 = process.hrtime.bigint();
      const outs = await Promise.all(xs.map((x, i) => sdk.evaluate(op, x, ys[i], bits)));
      const ms = Number(process.hrtime.bigint() - t0) / 1e6;
      clearInterval(timer);
      const vals = await Promise.all(outs.map((o) => sdk.decrypt(o)));
      ok = ok && vals.every((v, i) => v === model(vx(i), vy(i)));
      if (ms < all) { all = ms; ticks = turned; }
    }
    return { single_ms: best, all_ms: all, n, ticks, ok };
  };
  // an addition's bootstrap layers hold a few dozen ciphertexts: sixteen of them still fit the launches of one
  res.concurrent = await conc('add', 16, 16, (i) => 1000n * BigInt(i + 1), (i) => 77n + BigInt(i), (x, y) => (x + y) % (1n << 16n));
  // sixteen euint64 products are 29 000 bootstraps: they run at the batch rate instead of one after the other
  res.concurrentMul = await conc('mul', 64, 16, (i) => 1000003n * BigInt(i + 1), (i) => 77n + BigInt(i), (x, y) => (x * y) % (1n << 64n));
  // a failing request rejects its own promise only
  const mixed = await Promise.all([sdk.evaluate('add', a, b, 8), sdk.evaluate('add', a, Buffer.alloc(a.length), 8).then(() => 'resolved', () => 'rejected')]);
  res.mixed = [String(await sdk.decrypt(mixed[0])), mixed[1]];
  // ciphertexts resident on the GPU under bytes32 handles (the executor events' way,
  // tests/fhevm-suite/e2e/test/utils.ts:182-214): a chain le -> select -> sub stays on the device
  const H = (k) => Buffer.alloc(32, k);
  engine.storePut(H(1), a);
  engine.storePut(H(2), b);
  engine.evaluateBatchHandles('le', 8, [H(2)], [H(1)], null, [H(3)]);            // 17 <= 42
  engine.evaluateBatchHandles('select', 8, [H(2)], [H(1)], [H(3)], [H(4)]);       // true ? 17 : 42
  engine.evaluateBatchHandles('sub', 8, [H(1)], [H(4)], null, [H(5)]);            // 42 - 17
  engine.evaluateBatchHandles('add', 8, [H(5)], [100], null, [H(5)]);             // clear operand, result replaces its operand
  const stat = engine.storeStat(H(5));
  res.store = { chain: String(engine.decrypt(engine.storeGet(H(5)))), cond: String(engine.decrypt(engine.storeGet(H(3)))),
    entries: stat.entries, bitWidth: stat.bitWidth, condWidth: engine.storeStat(H(3)).bitWidth };
  engine.storeDrop(H(4));
  res.store.afterDrop = engine.storeStat(null).entries;
  res.store.dropped = engine.storeStat(H(4)).bitWidth;
  try { engine.evaluateBatchHandles('add', 8, [H(4)], [H(1)], null, [H(6)]); res.store.unknown = 'accepted'; } catch (e) { res.store.unknown = 'rejected'; }
  // compression for storage: both operands into one blob and back
  engine.compressionKeygen();
  const blob = engine.compress([a, b]);
  const back = engine.decompress(blob);
  res.blobBytes = blob.length;
  res.restored = [String(engine.decrypt(back.slice(0, a.length))), String(engine.decrypt(back.slice(a.length)))];
  // a client's compact input list, when the caller prepared one (a JSON file: base64 of 64-bit words):
  // keys made elsewhere are imported, the list is expanded, cast to the compute key and decrypted
  let compactOk = true;
  if (process.env.LFHE_COMPACT_CASE) {
    const c = JSON.parse(require('fs').readFileSync(process.env.LFHE_COMPACT_CASE, 'utf8'));
    const words = (b64) => { const b = Buffer.from(b64, 'base64'); return new BigUint64Array(b.buffer, b.byteOffset, b.length / 8); };
    const e2 = new NativeEngine({ seed: Buffer.alloc(32, 9), keygen: false });
    e2.importClientKeys(words(c.lweSk), words(c.glweSk));
    e2.setPkeKey(words(c.pkeSk));
    const cts = e2.expandCompactList(words(c.list), c.widths, c.packed);
    res.compact = cts.map((ct) => String(e2.decrypt(ct)));
    compactOk = res.compact.length === c.values.length && res.compact.every((v, i) => v === c.values[i]);
    e2.close();
  }
  console.log(JSON.stringify(res));
  const ok = compactOk && res.health === 'ok' && res.add === '59' && res.sub === '25' && res.lt === '0' && res.eq === '1' &&
    res.addScalar === '242' && res.shl === String((42 << 3) & 255) && res.mul64 === '18434227713315959936' &&
    res.batch[0] === '1' && res.batch[1] === '0' && Number(res.rand) < 1024 &&
    res.concurrent.ok && res.concurrent.all_ms < 2 * res.concurrent.single_ms && res.concurrent.ticks > 2 &&
    res.concurrentMul.ok && res.concurrentMul.all_ms < 13 * res.concurrentMul.single_ms &&
    res.mixed[0] === '59' && res.mixed[1] === 'rejected' &&
    res.store.chain === '125' && res.store.cond === '1' && res.store.entries === 5 && res.store.bitWidth === 8 &&
    res.store.condWidth === 1 && res.store.afterDrop === 4 && res.store.dropped === 0 && res.store.unknown === 'rejected' &&
    res.restored[0] === '42' && res.restored[1] === '17' && back.length === a.length + b.length && blob.length < a.length / 8;
  engine.close();
  process.exit(ok ? 0 : 1);
})().catch((e) => { console.error(e.message); process.exit(2); });
