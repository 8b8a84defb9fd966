'use strict';
/*
 * In-process replacement for the HTTP engine the reference's SDK talks to.
 *
 * NativeEngine wraps the N-API addon (lfhe_napi.node -> liblfhe.so).  LuxFHEClient keeps the method
 * names of the reference class (packages/luxfhejs/src/index.ts:42-141: initialize, getPublicKey,
 * encrypt_uint8 ... encrypt_uint256, encrypt_address) and of the hardhat plugin
 * (packages/hardhat-plugin/src/index.ts:57-87: encrypt, decrypt, getPublicKey), and adds
 * evaluate(), which the reference only reaches over POST /evaluate (e2e/test/fhe.test.ts:105-175).
 * Unlike the reference's encrypt_uint64/128/256, BigInt values are not squeezed through Number().
 */
const path = require('path');
const crypto = require('crypto');
const native = require('./lfhe_napi.node');

const PARAMS = { Q64: 1, PRIME_BSK: 2, PRIME_BSK_GROUPED: 3, Q64_GROUPED: 4, PRIME_BSK_GROUPED3: 5, Q64_GROUPED3: 6 };
const KEY = { CLIENT: 1, SERVER: 2 };
const COMPARISONS = new Set(['eq', 'ne', 'ge', 'gt', 'le', 'lt']);
const WIDTHS = new Set([1, 8, 16, 32, 64, 128, 160, 256]);

function toLE32(value) {
  let v = BigInt(value);
  if (v < 0n) throw new RangeError('value must be non-negative');
  const out = Buffer.alloc(32);
  for (let i = 0; i < 32; i++) { out[i] = Number(v & 0xffn); v >>= 8n; }
  if (v !== 0n) throw new RangeError('value does not fit 256 bits');
  return out;
}

function fromLE32(buf) {
  let v = 0n;
  for (let i = 31; i >= 0; i--) v = (v << 8n) | BigInt(buf[i]);
  return v;
}

let opened = false;
function openLibrary(libPath) {
  if (opened) return;
  const p = libPath || process.env.LFHE_LIB || path.join(__dirname, '..', '..', 'fhe_amd', 'liblfhe.so');
  native.open(p);
  opened = true;
}

// evaluateAsync requests gathered over one turn of the event loop: one native call per group of a shape
function sendSingle(engine, it) {
  native.evaluateAsync(engine.handle, it.code, it.bitWidth, it.left, it.rhs, it.scalar, it.cond).then(it.resolve, it.reject);
}

function flushPending(engine) {
  const groups = engine._pending;
  engine._pending = new Map();
  engine._flushing = false;
  for (const items of groups.values()) {
    if (items.length === 1 || items[0].op === 'cast') { items.forEach((it) => sendSingle(engine, it)); continue; }
    const first = items[0];
    const outBits = COMPARISONS.has(first.op) ? 1 : first.bitWidth;
    let p;
    try {
      p = native.evaluateBatchAsync(engine.handle, first.code, first.bitWidth, items.length, Buffer.concat(items.map((it) => it.left)),
        first.rhs ? Buffer.concat(items.map((it) => it.rhs)) : null, first.scalar,
        first.cond ? Buffer.concat(items.map((it) => it.cond)) : null, outBits);
    } catch (e) {
      items.forEach((it) => sendSingle(engine, it));       // one of them is malformed: each gets its own verdict
      continue;
    }
    p.then((all) => {
      const sz = all.length / items.length;
      items.forEach((it, i) => it.resolve(all.slice(i * sz, (i + 1) * sz)));
    }, () => items.forEach((it) => sendSingle(engine, it)));
  }
}

class NativeEngine {
  /**
   * @param {{seed?: Uint8Array, device?: number, params?: number, libPath?: string, keygen?: boolean}} opts
   * seed: 32 bytes the keys derive from -- as secret as the keys, so the default is fresh
   * randomness; a fixed seed is for tests.  params: default PARAMS.PRIME_BSK_GROUPED3, as the
   * HTTP server and the Python binding's GpuBackend.
   */
  constructor(opts) {
    const o = opts || {};
    openLibrary(o.libPath);
    const seed = o.seed ? Buffer.from(o.seed) : crypto.randomBytes(32);
    if (seed.length !== 32) throw new RangeError('seed must be 32 bytes');
    this.handle = native.createContext(seed, o.device || 0, o.params || PARAMS.PRIME_BSK_GROUPED3);
    if (o.keygen !== false) native.keygen(this.handle);
  }

  health() { return JSON.parse(native.health(this.handle)); }
  publicKey() { return native.publicKey(this.handle); }
  exportKeys(which) { return native.keysExport(this.handle, which); }
  importKeys(which, blob) { native.keysImport(this.handle, which, Buffer.from(blob)); }

  encrypt(value, bitWidth) {
    if (!WIDTHS.has(bitWidth)) throw new Error(`Encryption failed: unsupported bitWidth ${bitWidth}`);
    const m = (1n << BigInt(bitWidth)) - 1n;
    return native.encrypt(this.handle, toLE32(BigInt(value) & m), bitWidth, 0);
  }

  trivialEncrypt(value, bitWidth) {
    const m = (1n << BigInt(bitWidth)) - 1n;
    return native.encrypt(this.handle, toLE32(BigInt(value) & m), bitWidth, 1);
  }

  encryptPublic(publicKey, value, bitWidth) {
    const m = (1n << BigInt(bitWidth)) - 1n;
    return native.encryptPublic(this.handle, Buffer.from(publicKey), toLE32(BigInt(value) & m), bitWidth);
  }

  /** @returns {bigint} */
  decrypt(ciphertext) { return fromLE32(native.decrypt(this.handle, Buffer.from(ciphertext)).value); }

  /**
   * right: ciphertext bytes, or a number / bigint / decimal string for a clear operand.
   * cond: ebool ciphertext, only for op 'select'.
   */
  evaluate(op, bitWidth, left, right, cond) {
    const code = native.opFromName(op);
    if (code < 0) throw new Error(`unknown operation ${op}`);
    const scalar = typeof right === 'number' || typeof right === 'bigint' || typeof right === 'string';
    const rhs = right === undefined || right === null ? null : scalar ? toLE32(right) : Buffer.from(right);
    return native.evaluate(this.handle, code, bitWidth, Buffer.from(left), rhs, scalar, cond ? Buffer.from(cond) : null);
  }

  /**
   * evaluate() off the event loop: resolves with the result ciphertext.  Requests that are pending
   * together and agree in operator, width and operand kind are bootstrapped as one batch, so
   * `Promise.all` over many of them costs little more than one (the reference's tests fan out
   * this way: tests/fhevm-suite/e2e/test/instance.ts:32-36).
   * @returns {Promise<Buffer>}
   */
  evaluateAsync(op, bitWidth, left, right, cond) {
    const code = native.opFromName(op);
    if (code < 0) return Promise.reject(new Error(`unknown operation ${op}`));
    const scalar = typeof right === 'number' || typeof right === 'bigint' || typeof right === 'string';
    let item;
    try {
      const rhs = right === undefined || right === null ? null : scalar ? toLE32(right) : Buffer.from(right);
      item = { code, op, bitWidth, left: Buffer.from(left), rhs, scalar, cond: cond ? Buffer.from(cond) : null };
    } catch (e) {
      return Promise.reject(e);
    }
    // the calls of one turn of the event loop (a Promise.all over many) are gathered and sent together
    const key = `${code}:${bitWidth}:${scalar ? 's' : item.rhs ? 'c' : 'n'}:${item.cond ? 1 : 0}:${item.left.length}:${item.rhs ? item.rhs.length : 0}`;
    if (!this._pending) this._pending = new Map();
    const promise = new Promise((resolve, reject) => { item.resolve = resolve; item.reject = reject; });
    if (!this._pending.has(key)) this._pending.set(key, []);
    this._pending.get(key).push(item);
    if (!this._flushing) { this._flushing = true; setImmediate(() => flushPending(this)); }
    return promise;
  }

  /** n independent instances bootstrapped together; operands are arrays of ciphertexts (or of clear values). */
  evaluateBatch(op, bitWidth, lefts, rights, conds) {
    const code = native.opFromName(op);
    if (code < 0) throw new Error(`unknown operation ${op}`);
    const n = lefts.length;
    if (n === 0) return [];
    const scalar = rights && rights.length && (typeof rights[0] !== 'object');
    const L = Buffer.concat(lefts.map((b) => Buffer.from(b)));
    const R = rights ? Buffer.concat(rights.map((r) => (scalar ? toLE32(r) : Buffer.from(r)))) : null;
    const C = conds ? Buffer.concat(conds.map((b) => Buffer.from(b))) : null;
    const outBits = COMPARISONS.has(op) ? 1 : bitWidth;
    const all = native.evaluateBatch(this.handle, code, bitWidth, n, L, R, !!scalar, C, outBits);
    const sz = all.length / n;
    const out = [];
    for (let i = 0; i < n; i++) out.push(all.slice(i * sz, (i + 1) * sz));
    return out;
  }

  // ---- ciphertexts resident on the GPU under 32-byte handles (the executor events' bytes32 handles):
  // operands and results of a chain of operations stay in device memory, only storePut / storeGet cross PCIe
  storePut(handle, ciphertext) { native.storePut(this.handle, Buffer.from(handle), Buffer.from(ciphertext)); }
  storeGet(handle) { return native.storeGet(this.handle, Buffer.from(handle)); }
  storeDrop(handle) { native.storeDrop(this.handle, Buffer.from(handle)); }
  /** { entries, deviceBytes, bitWidth } -- bitWidth of `handle` (0: no such entry) when one is given. */
  storeStat(handle) { return native.storeStat(this.handle, handle ? Buffer.from(handle) : null); }

  /** evaluateBatch on stored ciphertexts: arrays of 32-byte handles in (rights: handles or clear values), results entered under outHandles. */
  evaluateBatchHandles(op, bitWidth, leftHandles, rights, condHandles, outHandles) {
    const code = native.opFromName(op);
    if (code < 0) throw new Error(`unknown operation ${op}`);
    const n = leftHandles.length;
    if (n === 0) return;
    if (outHandles.length !== n) throw new RangeError('one result handle per instance');
    const scalar = rights && rights.length && (typeof rights[0] !== 'object');
    const cat = (hs) => Buffer.concat(hs.map((h) => Buffer.from(h)));
    native.evaluateBatchHandles(this.handle, code, bitWidth, n, cat(leftHandles),
      rights ? Buffer.concat(rights.map((r) => (scalar ? toLE32(r) : Buffer.from(r)))) : null, !!scalar,
      condHandles ? cat(condHandles) : null, cat(outHandles));
  }

  /** Oblivious pseudo-random ciphertext (FheRand / FheRandBounded): uniform below 2^randomBits. */
  rand(bitWidth, seed16, randomBits) {
    return native.rand(this.handle, bitWidth, Buffer.from(seed16), randomBits || bitWidth);
  }

  /** Secret keys made elsewhere (arrays of 0/1 as BigUint64Array), then server keys for them. */
  importClientKeys(lweSk, glweSk) {
    native.importClientKeys(this.handle, Buffer.from(lweSk.buffer, lweSk.byteOffset, lweSk.byteLength),
      Buffer.from(glweSk.buffer, glweSk.byteOffset, glweSk.byteLength));
    native.keygenServer(this.handle);
  }

  /** Inputs encrypted under a separate public-key secret: set that key once, then cast ciphertexts. */
  setPkeKey(pkeSk) { native.setPkeKey(this.handle, Buffer.from(pkeSk.buffer, pkeSk.byteOffset, pkeSk.byteLength)); }
  castFromPke(ciphertext) { return native.castFromPke(this.handle, Buffer.from(ciphertext)); }

  /**
   * A client's compact ciphertext list (mask of 2048 words, then one body word per ciphertext, as a
   * BigUint64Array) -> one ciphertext per entry of `widths` under the compute key; packed: two radix
   * blocks per list ciphertext (build_with_proof_packed).  Needs setPkeKey first.
   * @returns {Buffer[]}
   */
  expandCompactList(words, widths, packed) {
    const w = Uint32Array.from(widths);
    const all = native.expandCompactList(this.handle, Buffer.from(words.buffer, words.byteOffset, words.byteLength),
      packed !== false, Buffer.from(w.buffer));
    const out = [];
    let off = 0;
    for (const bits of widths) {
      const size = 16 + (bits === 1 ? 1 : bits / 2) * 2049 * 8;
      out.push(all.slice(off, off + size));
      off += size;
    }
    return out;
  }

  /** Compression for storage: keys once, then any list of ciphertexts <-> one blob (one bootstrap per block on the way back). */
  compressionKeygen() { native.compressionKeygen(this.handle); }
  compress(ciphertexts) { return native.compress(this.handle, Buffer.concat(ciphertexts.map((c) => Buffer.from(c)))); }
  /** @returns {Buffer} the ciphertexts back to back, in the order they were given to compress */
  decompress(blob) { return native.decompress(this.handle, Buffer.from(blob)); }

  pbsCount() { return native.pbsCount(this.handle); }
  close() { if (this.handle) { native.destroy(this.handle); this.handle = null; } }
}

class LuxFHEClient {
  /** @param {{engine?: NativeEngine, provider?: any, seed?: Uint8Array, device?: number}} config */
  constructor(config) {
    const c = config || {};
    this.provider = c.provider;
    this.engine = c.engine || new NativeEngine({ seed: c.seed, device: c.device });
    this._publicKey = null;
  }

  async initialize() { await this.getPublicKey(); }

  async getPublicKey() {
    if (!this._publicKey) this._publicKey = new Uint8Array(this.engine.publicKey());
    return this._publicKey;
  }

  async encrypt_uint8(value) { return this.encrypt(value, 8); }
  async encrypt_uint16(value) { return this.encrypt(value, 16); }
  async encrypt_uint32(value) { return this.encrypt(value, 32); }
  async encrypt_uint64(value) { return this.encrypt(value, 64); }
  async encrypt_uint128(value) { return this.encrypt(value, 128); }
  async encrypt_uint256(value) { return this.encrypt(value, 256); }
  async encrypt_address(address) { return this.encrypt(BigInt(address), 160); }
  async encrypt_bool(value) { return this.encrypt(value ? 1 : 0, 1); }

  async encrypt(value, bitWidth) { return new Uint8Array(this.engine.encrypt(value, bitWidth)); }
  async decrypt(ciphertext) { return this.engine.decrypt(ciphertext); }
  /** Runs on a worker thread; concurrent calls of one shape share their bootstrap launches. */
  async evaluate(op, left, right, bitWidth, cond) {
    return new Uint8Array(await this.engine.evaluateAsync(op, bitWidth, left, right, cond));
  }
}

module.exports = { NativeEngine, LuxFHEClient, PARAMS, KEY, toLE32, fromLE32, native };
