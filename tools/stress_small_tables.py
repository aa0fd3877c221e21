"""Build variants of the product library whose rarely taken paths run all the time, then stress them against the oracle.

    python tools/stress_small_tables.py [N]              # kmcex_amd/libkmx_smalltab.so: finisher LDS tables of 2^N slots
                                                         # (-DKMX_FIN_LOG2=N, default 5): slot sharing, table-2 route,
                                                         # foreign MARK words and index ranges in every round
    python tools/stress_small_tables.py --cl-cap C       # kmcex_amd/libkmx_smallcap.so: claim bins of C tuples
                                                         # (-DKMX_CL_CAP=C): bins overflow, whole lists take the ordered path
    (both build here, no GPU needed)
    KMX_LIBRARY=kmcex_amd/libkmx_smalltab.so python tools/stress_parity.py 300            # on the GPU box
    KMX_LIBRARY=kmcex_amd/libkmx_smallcap.so KMX_BS_CAP=64 python tools/stress_parity.py 200 5 small
"""
import os as _os; _os.environ.setdefault("KMX_TEST_HOOKS", "1")   # forced code paths are test hooks
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "kmcex_amd/csrc")
inc = "-I" + os.path.join(root, "include")
if "--cl-cap" in sys.argv:
    cap = int(sys.argv[sys.argv.index("--cl-cap") + 1])
    objs = []
    for src in ("kernels.hip", "kmx_api.hip"):                   # the capacity sizes kernels and allocation alike
        obj = os.path.join(d, src.replace(".hip", "_smallcap.o"))
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DKMX_CL_CAP={cap}", inc, "-c", os.path.join(d, src), "-o", obj])
        objs.append(obj)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(root, "kmcex_amd/libkmx_smallcap.so")] + objs + [os.path.join(d, f) for f in ("rest_device.o", "kmc_reader.o")])
    for o in objs:
        os.remove(o)
    print(f"built kmcex_amd/libkmx_smallcap.so with {cap} tuples per claim bin")
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    obj = os.path.join(d, "kernels_smalltab.o")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", f"-DKMX_FIN_LOG2={n}", inc, "-c", os.path.join(d, "kernels.hip"), "-o", obj])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(root, "kmcex_amd/libkmx_smalltab.so"), obj] + [os.path.join(d, f) for f in ("rest_device.o", "kmx_api.o", "kmc_reader.o")])
    os.remove(obj)
    print(f"built kmcex_amd/libkmx_smalltab.so with 2^{n} slots per finisher table")
