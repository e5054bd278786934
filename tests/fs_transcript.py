"""Fiat-Shamir transcript of the reference, restated for the test harness (the transcript is host
code that STAYS with the caller in an integration; the GPU never sees it).

Reference: lib/random/transcript.h:33-190 (SHA-256 running state; reads = AES-256-ECB counter PRF keyed by
SHA-256 of a copy of the state), lib/sumcheck/transcript_sumcheck.h:31-82, lib/random/random.h:57-105.
Only hashlib is available here, so AES-256 (FIPS-197) is implemented below; it is pinned by the FIPS-197
C.3 known-answer vector in tests/test_sumcheck_e2e.py.
"""
import hashlib
import struct

_SBOX = None


def _init_sbox():
    global _SBOX
    if _SBOX is not None:
        return
    # multiplicative inverse in GF(2^8) + affine map
    exp, log = [0] * 512, [0] * 256
    x = 1
    for i in range(255):
        exp[i] = x
        log[x] = i
        x ^= (x << 1) ^ (0x11B if x & 0x80 else 0)
        x &= 0xFF
    for i in range(255, 512):
        exp[i] = exp[i - 255]
    sb = [0] * 256
    for a in range(256):
        inv = 0 if a == 0 else exp[255 - log[a]]
        s = inv
        for sh in (1, 2, 3, 4):
            s ^= ((inv << sh) | (inv >> (8 - sh))) & 0xFF
        sb[a] = s ^ 0x63
    _SBOX = sb


def _xt(a):
    return ((a << 1) ^ 0x1B) & 0xFF if a & 0x80 else a << 1


class AES256:
    def __init__(self, key):
        _init_sbox()
        assert len(key) == 32
        w = [list(key[4 * i:4 * i + 4]) for i in range(8)]
        rcon = 1
        for i in range(8, 60):
            t = list(w[i - 1])
            if i % 8 == 0:
                t = t[1:] + t[:1]
                t = [_SBOX[b] for b in t]
                t[0] ^= rcon
                rcon = _xt(rcon)
            elif i % 8 == 4:
                t = [_SBOX[b] for b in t]
            w.append([a ^ b for a, b in zip(w[i - 8], t)])
        self.rk = [sum((w[4 * r + c] for c in range(4)), []) for r in range(15)]

    def encrypt_block(self, block):
        s = [b ^ k for b, k in zip(block, self.rk[0])]
        for rnd in range(1, 15):
            s = [_SBOX[b] for b in s]
            # shift rows (state is column-major: s[4*c + r])
            s = [s[4 * ((c + r) % 4) + r] for c in range(4) for r in range(4)]
            if rnd != 14:
                t = []
                for c in range(4):
                    a = s[4 * c:4 * c + 4]
                    x = a[0] ^ a[1] ^ a[2] ^ a[3]
                    t += [a[0] ^ x ^ _xt(a[0] ^ a[1]), a[1] ^ x ^ _xt(a[1] ^ a[2]), a[2] ^ x ^ _xt(a[2] ^ a[3]),
                          a[3] ^ x ^ _xt(a[3] ^ a[0])]
                s = t
            s = [b ^ k for b, k in zip(s, self.rk[rnd])]
        return bytes(s)


class _OpenSslAes256:
    """AES-256-ECB through the system libcrypto (what the reference itself calls, lib/util/crypto.h:74-103);
    used when loadable because the pure-Python cipher above costs ~0.3 ms per block."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            import ctypes
            import ctypes.util
            name = ctypes.util.find_library("crypto")
            L = ctypes.CDLL(name) if name else None
            if L is None:
                raise OSError("libcrypto not found")
            L.EVP_CIPHER_CTX_new.restype = ctypes.c_void_p
            L.EVP_aes_256_ecb.restype = ctypes.c_void_p
            L.EVP_EncryptInit_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
            L.EVP_EncryptUpdate.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_int]
            L.EVP_CIPHER_CTX_free.argtypes = [ctypes.c_void_p]
            cls._lib = L
        return cls._lib

    def __init__(self, key):
        import ctypes
        L = self.lib()
        self.ctx = L.EVP_CIPHER_CTX_new()
        assert L.EVP_EncryptInit_ex(self.ctx, L.EVP_aes_256_ecb(), None, bytes(key), None) == 1
        self._out = ctypes.create_string_buffer(32)
        self._n = ctypes.c_int(0)

    def encrypt_block(self, block):
        import ctypes
        L = self.lib()
        assert L.EVP_EncryptUpdate(self.ctx, self._out, ctypes.byref(self._n), bytes(block), 16) == 1
        return self._out.raw[:16]

    def __del__(self):
        try:
            self.lib().EVP_CIPHER_CTX_free(self.ctx)
        except Exception:
            pass


def make_aes256(key):
    try:
        return _OpenSslAes256(key)
    except Exception:
        return AES256(key)


class Transcript:
    """lib/random/transcript.h:70-190"""

    def __init__(self, init):
        self.sha = hashlib.sha256()
        self.prf = None
        self.write_bytes(init)

    def clone(self):
        t = Transcript.__new__(Transcript)
        t.sha = self.sha.copy()
        t.prf = None
        return t

    def _upd(self, data):
        self.prf = None  # any write invalidates the PRF (:174-178)
        self.sha.update(data)

    def write_bytes(self, data):  # tag 0 || u64 length || bytes (:115-120)
        self._upd(b"\x00" + struct.pack("<Q", len(data)) + bytes(data))

    def write_elt(self, e16):  # tag 1 || to_bytes_field (:136-140)
        self._upd(b"\x01" + bytes(e16))

    def write_array(self, elts):  # tag 2 || u64 count || elements (:144-152)
        self._upd(b"\x02" + struct.pack("<Q", len(elts)) + b"".join(bytes(e) for e in elts))

    def bytes(self, n):
        if self.prf is None:
            self.prf = [make_aes256(self.sha.copy().digest()), 0, b"", 0]  # cipher, next block, saved, read ptr
        out = bytearray()
        while len(out) < n:
            c, nb, saved, rp = self.prf
            if rp == len(saved):
                saved = c.encrypt_block(struct.pack("<Q", nb) + b"\x00" * 8)  # FSPRF::refill (:53-60)
                self.prf = [c, nb + 1, saved, 0]
                rp = 0
            take = min(n - len(out), 16 - rp)
            out += saved[rp:rp + take]
            self.prf[3] = rp + take
        return bytes(out)

    def elt_gf2128(self):  # GF2_128::sample: 16 bytes LE (lib/gf2k/gf2_128.h:182-190)
        return self.bytes(16)

    # RandomEngine::nat / choose (lib/random/random.h:57-105)
    def nat(self, n):
        assert n > 0
        l, nn = 0, n
        while nn:
            nn >>= 8
            l += 1
        mask = 0
        while (n & mask) != n:
            mask = (mask << 1) | 1
        while True:
            r = int.from_bytes(self.bytes(l), "little") & mask
            if r < n:
                return r

    def choose(self, n, k):
        A = list(range(n))
        res = []
        for i in range(k):
            j = i + self.nat(n - i)
            A[i], A[j] = A[j], A[i]
            res.append(A[i])
        return res
