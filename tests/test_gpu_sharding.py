"""What a rank other than 0 does in a multi-GPU run, on one GPU: it receives the two device key
images (here: a device-to-device copy stands in for the RCCL broadcast), attaches them and must
then bootstrap exactly like the context that generated the keys."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("params_id", [1, 2, 3, 4, 5, 6],
                         ids=["bsk_q64", "bsk_prime", "bsk_prime_grouped", "bsk_q64_grouped", "bsk_prime_grouped3", "bsk_q64_grouped3"])
def test_attached_key_images_bootstrap_like_the_originals(params_id):
    import ctypes as C
    from fhe_amd.native import Context, KEY_CLIENT
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    D2D = 3
    src = Context(seed=bytes(range(32)), params_id=params_id)
    dst = Context(seed=bytes(range(32)), params_id=params_id)
    tb, tk = C.c_void_p(), C.c_void_p()
    try:
        src.keygen()
        bp, bsk_bytes, kp, ksk_bytes = src.device_keys()
        assert (bsk_bytes, ksk_bytes) == dst.device_keys_size()
        assert hip.hipMalloc(C.byref(tb), bsk_bytes) == 0 and hip.hipMalloc(C.byref(tk), ksk_bytes) == 0
        assert hip.hipMemcpy(tb, C.c_void_p(bp), bsk_bytes, D2D) == 0      # synchronous
        assert hip.hipMemcpy(tk, C.c_void_p(kp), ksk_bytes, D2D) == 0
        dst.attach_device_keys(tb.value, tk.value)
        dst.import_keys(KEY_CLIENT, src.export_keys(KEY_CLIENT))
        rng = np.random.default_rng(3)
        vals = [int(v) for v in rng.integers(0, 256, size=80)]           # 320 blocks: the matrix-core key switch
        blocks = np.concatenate([np.frombuffer(src.encrypt(v, 8)[16:], dtype=np.uint64).reshape(4, 2049) for v in vals])
        luts = np.stack([src.make_lut(lambda x: (3 * x + 1) % 16), src.make_lut(lambda x: x)])
        idx = (np.arange(len(blocks)) % 2).astype(np.uint32)
        a, b = src.pbs_batch(blocks, luts, idx), dst.pbs_batch(blocks, luts, idx)
        assert np.array_equal(a, b)
        # and the operator level on the attached context
        x, y = dst.encrypt(200, 8), dst.encrypt(77, 8)
        assert dst.decrypt(dst.evaluate("add", 8, x, y)) == (200 + 77) & 0xFF
        assert src.decrypt(dst.evaluate("mul", 8, x, y)) == (200 * 77) & 0xFF
    finally:
        dst.close()
        src.close()
        hip.hipFree(tb)
        hip.hipFree(tk)
