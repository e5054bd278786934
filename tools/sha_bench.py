"""Host SHA-256 of the library's built-in transcript (csrc/fs_crypto.cc) next to OpenSSL's: the Fiat-Shamir preamble of a proof is a
SHA-256 over `nterms` zero bytes (zk_common.h:163-180) -- 7.76 MB for the mdoc hash circuit -- sequential host work on the critical
path of every proof.  EPYC 9575F (SHA-NI): 2.44 GB/s against 2.12 for OpenSSL, i.e. 3.2 ms of the mdoc hash proof and 2.0 ms of flatsha-32."""
import ctypes as C, time, hashlib, sys
sys.path.insert(0,"/root/repo")
import __graft_entry__ as ge
L=C.CDLL(ge.LIB)
L.lfgpu_sha256.argtypes=[C.c_void_p,C.c_size_t,C.c_void_p]
L.lfgpu_crypto_hw.restype=C.c_int
print("crypto hw flags", L.lfgpu_crypto_hw())
n=8*1024*1024
buf=(C.c_uint8*n)()
out=(C.c_uint8*32)()
for rep in range(3):
    t=time.perf_counter(); L.lfgpu_sha256(buf,n,out); dt=time.perf_counter()-t
    print("lfgpu_sha256 %.2f ms  %.2f GB/s"%(dt*1e3, n/dt/1e9))
t=time.perf_counter(); h=hashlib.sha256(bytes(n)).digest(); dt=time.perf_counter()-t
print("hashlib(openssl) %.2f ms %.2f GB/s"%(dt*1e3,n/dt/1e9), bytes(out)==h)
