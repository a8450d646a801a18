"""Worker of tests/test_rendezvous_cpu.py: one rank of the communicator bootstrap (cognn_rccl_rendezvous_tcp); prints the id it ends up with."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    rank, world, port, timeout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    from cognn_amd import capi
    lib = capi.load()
    ident = (ctypes.c_uint8 * 128)()
    rc = lib.cognn_rccl_rendezvous_tcp(b"127.0.0.1", port, rank, world, timeout, ident)
    if rc != 0:
        print("ERROR " + lib.cognn_exchange_last_error().decode())
        sys.exit(3)
    print("ID " + bytes(ident).hex())
