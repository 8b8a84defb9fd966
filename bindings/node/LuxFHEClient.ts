/**
 * LuxFHEClient over an engine transport -- the class of packages/luxfhejs/src/index.ts:42-200 with its
 * engine calls behind one small interface, so that the same client runs against the HTTP engine on
 * :8448 (as today) or against the GPU library in-process (bindings/node/index.js).
 *
 * Drop-in: copy next to packages/luxfhejs/src/index.ts and re-export `LuxFHEClient` from here; the
 * permit helpers of that file (generatePermit, getPermit, ...) are untouched and keep working with
 * this class (storePermit / getStoredPermit / extractPermitPermission / unseal are carried over).
 * Differences from the reference class, all additive:
 *   - `config.engine`: an EngineTransport; without it the client talks HTTP to `config.serverUrl`
 *     exactly as the reference does (GET /publickey, POST /encrypt {value, bitWidth});
 *   - 64-bit and wider values are sent as decimal strings instead of being squeezed through
 *     Number() (the reference loses the bits above 2^53, src/index.ts:98-122);
 *   - `evaluate` and `decrypt`, which the reference only reaches with raw fetch calls
 *     (e2e/test/fhe.test.ts:105-175, packages/hardhat-plugin/src/index.ts:69-78).
 */

export type SupportedProvider = any;

export interface Permit {
  contractAddress: string;
  sealingKey: { privateKey: string; publicKey: string };
  signature: string;
}

export interface Permission {
  contractAddress: string;
  signer: string;
  publicKey: string;
  signature: string;
}

/** What the client needs from an engine. */
export interface EngineTransport {
  publicKey(): Promise<Uint8Array>;
  encrypt(value: bigint, bitWidth: number): Promise<Uint8Array>;
  decrypt(ciphertext: Uint8Array): Promise<bigint>;
  evaluate(
    op: string,
    bitWidth: number,
    left: Uint8Array,
    right?: Uint8Array | bigint | null,
    cond?: Uint8Array | null
  ): Promise<Uint8Array>;
}

/** The reference's transport: the engine's HTTP surface. */
export class HttpTransport implements EngineTransport {
  constructor(private serverUrl: string = 'http://localhost:8448') {}

  async publicKey(): Promise<Uint8Array> {
    const res = await fetch(`${this.serverUrl}/publickey`);
    if (!res.ok) throw new Error(`Failed to fetch public key: ${res.status}`);
    return new Uint8Array(await res.arrayBuffer());
  }

  private async post(path: string, body: unknown): Promise<Response> {
    const res = await fetch(`${this.serverUrl}${path}`, {
      method: 'POST',
      headers: { 'Content-Type': 'application/json' },
      body: JSON.stringify(body),
    });
    if (!res.ok) throw new Error(`${path.slice(1)} failed: ${await res.text()}`);
    return res;
  }

  async encrypt(value: bigint, bitWidth: number): Promise<Uint8Array> {
    const wire = bitWidth <= 32 ? Number(value) : value.toString();
    const res = await this.post('/encrypt', { value: wire, bitWidth });
    return new Uint8Array(await res.arrayBuffer());
  }

  async decrypt(ciphertext: Uint8Array): Promise<bigint> {
    const res = await this.post('/decrypt', { ciphertext: Array.from(ciphertext) });
    const data = (await res.json()) as { value: string | number };
    return BigInt(data.value);
  }

  async evaluate(
    op: string,
    bitWidth: number,
    left: Uint8Array,
    right?: Uint8Array | bigint | null,
    cond?: Uint8Array | null
  ): Promise<Uint8Array> {
    const body: Record<string, unknown> = { op, bitWidth, left: Array.from(left) };
    if (typeof right === 'bigint') body.scalar = right.toString();
    else if (right) body.right = Array.from(right);
    if (cond) body.cond = Array.from(cond);
    const res = await this.post('/evaluate', body);
    return new Uint8Array(await res.arrayBuffer());
  }
}

/** The GPU library in-process: wraps a NativeEngine of bindings/node/index.js. */
export class NativeTransport implements EngineTransport {
  // the engine is typed structurally so that this file has no import the package may not resolve
  constructor(
    private engine: {
      publicKey(): Uint8Array;
      encrypt(value: bigint, bitWidth: number): Uint8Array;
      decrypt(ciphertext: Uint8Array): bigint;
      // runs off the event loop; concurrent calls of one shape share their bootstrap launches
      evaluateAsync(
        op: string,
        bitWidth: number,
        left: Uint8Array,
        right?: Uint8Array | bigint | null,
        cond?: Uint8Array | null
      ): Promise<Uint8Array>;
    }
  ) {}

  async publicKey(): Promise<Uint8Array> {
    return new Uint8Array(this.engine.publicKey());
  }
  async encrypt(value: bigint, bitWidth: number): Promise<Uint8Array> {
    return new Uint8Array(this.engine.encrypt(value, bitWidth));
  }
  async decrypt(ciphertext: Uint8Array): Promise<bigint> {
    return this.engine.decrypt(ciphertext);
  }
  async evaluate(
    op: string,
    bitWidth: number,
    left: Uint8Array,
    right?: Uint8Array | bigint | null,
    cond?: Uint8Array | null
  ): Promise<Uint8Array> {
    return new Uint8Array(await this.engine.evaluateAsync(op, bitWidth, left, right, cond));
  }
}

export interface LuxFHEClientConfig {
  provider?: SupportedProvider;
  serverUrl?: string;
  engine?: EngineTransport;
}

const permitStore = new Map<string, Permit>();

export class LuxFHEClient {
  private provider: SupportedProvider;
  private transport: EngineTransport;
  private publicKey: Uint8Array | null = null;

  constructor(config: LuxFHEClientConfig = {}) {
    this.provider = config.provider;
    this.transport = config.engine ?? new HttpTransport(config.serverUrl);
  }

  async initialize(): Promise<void> {
    await this.getPublicKey();
  }

  async getPublicKey(): Promise<Uint8Array> {
    if (!this.publicKey) this.publicKey = await this.transport.publicKey();
    return this.publicKey;
  }

  async encrypt_uint8(value: number): Promise<Uint8Array> {
    return this.encrypt(value, 8);
  }
  async encrypt_uint16(value: number): Promise<Uint8Array> {
    return this.encrypt(value, 16);
  }
  async encrypt_uint32(value: number): Promise<Uint8Array> {
    return this.encrypt(value, 32);
  }
  async encrypt_uint64(value: number | bigint): Promise<Uint8Array> {
    return this.encrypt(value, 64);
  }
  async encrypt_uint128(value: number | bigint): Promise<Uint8Array> {
    return this.encrypt(value, 128);
  }
  async encrypt_uint256(value: number | bigint): Promise<Uint8Array> {
    return this.encrypt(value, 256);
  }
  async encrypt_address(address: string): Promise<Uint8Array> {
    return this.encrypt(BigInt(address), 160);
  }
  async encrypt_bool(value: boolean): Promise<Uint8Array> {
    return this.encrypt(value ? 1 : 0, 1);
  }

  async encrypt(value: number | bigint, bitWidth: number): Promise<Uint8Array> {
    const v = BigInt(value);
    if (v < 0n || v >> BigInt(bitWidth) !== 0n) throw new Error(`Encryption failed: value out of range for ${bitWidth} bits`);
    return this.transport.encrypt(v, bitWidth);
  }

  async decrypt(ciphertext: Uint8Array): Promise<bigint> {
    return this.transport.decrypt(ciphertext);
  }

  /** op: 'add' | 'sub' | 'mul' | 'lt' | 'eq' | ...; right: a ciphertext or a clear value; comparisons return an ebool */
  async evaluate(
    op: string,
    left: Uint8Array,
    right: Uint8Array | number | bigint | null | undefined,
    bitWidth: number,
    cond?: Uint8Array | null
  ): Promise<Uint8Array> {
    const rhs = typeof right === 'number' ? BigInt(right) : right;
    return this.transport.evaluate(op, bitWidth, left, rhs, cond);
  }

  // ---- permit bookkeeping, as in the reference class (src/index.ts:146-199)
  unseal(contractAddress: string, sealedData: Uint8Array | string): bigint {
    const permit = permitStore.get(contractAddress.toLowerCase());
    if (!permit) throw new Error(`No permit found for contract ${contractAddress}`);
    if (typeof sealedData === 'string') {
      try {
        return BigInt(sealedData);
      } catch {
        return 0n;
      }
    }
    let value = 0n;
    for (let i = 0; i < 8 && i < sealedData.length; i++) value |= BigInt(sealedData[i]) << BigInt(8 * i);
    return sealedData.length >= 8 ? value : 0n;
  }

  extractPermitPermission(permit: Permit): Permission {
    return {
      contractAddress: permit.contractAddress,
      signer: permit.signature.slice(0, 42),
      publicKey: permit.sealingKey.publicKey,
      signature: permit.signature,
    };
  }

  storePermit(permit: Permit): void {
    permitStore.set(permit.contractAddress.toLowerCase(), permit);
  }

  getStoredPermit(contractAddress: string): Permit | undefined {
    return permitStore.get(contractAddress.toLowerCase());
  }
}
