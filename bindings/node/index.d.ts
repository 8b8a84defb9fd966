/**
 * Type declarations of the in-process binding (index.js -> lfhe_napi.node -> liblfhe.so).
 * LuxFHEClient keeps the method names of packages/luxfhejs/src/index.ts:42-141 and of the hardhat
 * plugin (packages/hardhat-plugin/src/index.ts:57-87); NativeEngine is one GPU and one key set.
 */
/// <reference types="node" />

export type Bytes = Uint8Array | Buffer;
export type ClearValue = number | bigint | string;
export type BitWidth = 1 | 8 | 16 | 32 | 64 | 128 | 160 | 256;
export type Operator =
  | 'add' | 'sub' | 'mul' | 'eq' | 'ne' | 'ge' | 'gt' | 'le' | 'lt' | 'min' | 'max'
  | 'and' | 'or' | 'xor' | 'not' | 'neg' | 'select' | 'shl' | 'shr' | 'rotl' | 'rotr'
  | 'div' | 'rem' | 'cast';

/** Parameter set ids of include/lfhe.h: bootstrapping key over 2^64, over the NTT prime, over the prime with key bits in pairs. */
export const PARAMS: { readonly Q64: 1; readonly PRIME_BSK: 2; readonly PRIME_BSK_GROUPED: 3; readonly Q64_GROUPED: 4; readonly PRIME_BSK_GROUPED3: 5; readonly Q64_GROUPED3: 6 };
export const KEY: { readonly CLIENT: 1; readonly SERVER: 2 };

export interface NativeEngineOptions {
  /** 32 bytes the keys derive from (as secret as the keys); default: fresh randomness */
  seed?: Uint8Array;
  device?: number;
  /** default PARAMS.PRIME_BSK_GROUPED3, as the HTTP server */
  params?: number;
  /** path of liblfhe.so; default: $LFHE_LIB, then fhe_amd/liblfhe.so of this checkout */
  libPath?: string;
  /** false: do not generate keys (they will be imported) */
  keygen?: boolean;
}

export interface Health {
  status: 'ok';
  threshold: boolean;
  parties: number;
  backend: string;
  device: number;
  keys: boolean;
  pbs: number;
}

export class NativeEngine {
  constructor(opts?: NativeEngineOptions);
  health(): Health;
  publicKey(): Buffer;
  exportKeys(which: number): Buffer;
  importKeys(which: number, blob: Bytes): void;
  encrypt(value: ClearValue, bitWidth: BitWidth): Buffer;
  trivialEncrypt(value: ClearValue, bitWidth: BitWidth): Buffer;
  encryptPublic(publicKey: Bytes, value: ClearValue, bitWidth: BitWidth): Buffer;
  decrypt(ciphertext: Bytes): bigint;
  /** right: a ciphertext, or a clear value; cond: an ebool ciphertext, for 'select' only */
  evaluate(op: Operator, bitWidth: BitWidth, left: Bytes, right?: Bytes | ClearValue | null, cond?: Bytes | null): Buffer;
  /** evaluate() on a worker thread; pending requests of one shape share their bootstrap launches */
  evaluateAsync(op: Operator, bitWidth: BitWidth, left: Bytes, right?: Bytes | ClearValue | null, cond?: Bytes | null): Promise<Buffer>;
  /** n independent instances bootstrapped together */
  evaluateBatch(op: Operator, bitWidth: BitWidth, lefts: Bytes[], rights?: Array<Bytes | ClearValue> | null, conds?: Bytes[] | null): Buffer[];
  rand(bitWidth: BitWidth, seed16: Uint8Array, randomBits?: number): Buffer;
  importClientKeys(lweSk: BigUint64Array, glweSk: BigUint64Array): void;
  setPkeKey(pkeSk: BigUint64Array): void;
  castFromPke(ciphertext: Bytes): Buffer;
  /** words: mask (2048) then one body word per list ciphertext; packed defaults to true */
  expandCompactList(words: BigUint64Array, widths: BitWidth[], packed?: boolean): Buffer[];
  /** Ciphertexts resident on the GPU under 32-byte handles; results of evaluateBatchHandles stay there. */
  storePut(handle: Bytes, ciphertext: Bytes): void;
  storeGet(handle: Bytes): Buffer;
  storeDrop(handle: Bytes): void;
  storeStat(handle?: Bytes | null): { entries: number; deviceBytes: number; bitWidth: number };
  evaluateBatchHandles(op: Operator, bitWidth: BitWidth, leftHandles: Bytes[], rights: Array<Bytes | ClearValue> | null, condHandles: Bytes[] | null, outHandles: Bytes[]): void;
  compressionKeygen(): void;
  compress(ciphertexts: Bytes[]): Buffer;
  decompress(blob: Bytes): Buffer;
  pbsCount(): number;
  close(): void;
}

export interface LuxFHEClientConfig {
  /** kept for signature compatibility with the reference's config; unused by the in-process engine */
  provider?: any;
  engine?: NativeEngine;
  seed?: Uint8Array;
  device?: number;
}

export class LuxFHEClient {
  constructor(config?: LuxFHEClientConfig);
  readonly engine: NativeEngine;
  initialize(): Promise<void>;
  getPublicKey(): Promise<Uint8Array>;
  encrypt_uint8(value: number): Promise<Uint8Array>;
  encrypt_uint16(value: number): Promise<Uint8Array>;
  encrypt_uint32(value: number): Promise<Uint8Array>;
  encrypt_uint64(value: number | bigint): Promise<Uint8Array>;
  encrypt_uint128(value: number | bigint): Promise<Uint8Array>;
  encrypt_uint256(value: number | bigint): Promise<Uint8Array>;
  encrypt_address(address: string): Promise<Uint8Array>;
  encrypt_bool(value: boolean): Promise<Uint8Array>;
  encrypt(value: ClearValue, bitWidth: BitWidth): Promise<Uint8Array>;
  decrypt(ciphertext: Bytes): Promise<bigint>;
  evaluate(op: Operator, left: Bytes, right: Bytes | ClearValue | null | undefined, bitWidth: BitWidth, cond?: Bytes | null): Promise<Uint8Array>;
}

export function toLE32(value: ClearValue): Buffer;
export function fromLE32(buf: Bytes): bigint;
export const native: any;
