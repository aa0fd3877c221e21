"""Build kmcex_amd/libkmx_smalltab.so: the product library with the finisher's LDS tables shrunk to 2^N slots
(-DKMX_FIN_LOG2=N, default 5 = 32 slots), so that slot sharing, the table-2 route, foreign MARK words and index ranges
happen in every round instead of once in a while.  Then run the randomised parity stress against it:

    python tools/stress_small_tables.py [N]            # build (here, no GPU needed)
    KMX_LIBRARY=kmcex_amd/libkmx_smalltab.so python tools/stress_parity.py 300     # on the GPU box
"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
d = os.path.join(root, "kmcex_amd/csrc")
obj = os.path.join(d, "kernels_smalltab.o")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", f"-DKMX_FIN_LOG2={n}", "-I" + os.path.join(root, "include"), "-c", os.path.join(d, "kernels.hip"), "-o", obj])
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(root, "kmcex_amd/libkmx_smalltab.so"), obj] + [os.path.join(d, f) for f in ("rest_device.o", "kmx_api.o", "kmc_reader.o")])
os.remove(obj)
print(f"built kmcex_amd/libkmx_smalltab.so with 2^{n} slots per finisher table")
