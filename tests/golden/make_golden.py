#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref/ref_driver).

Run in the build container only (needs /root/reference to have been compiled by `make -C oracle ref`).
For every case below it
  1. builds a synthetic KMC1 database (kmcex_amd.synth + kmcex_amd.kmcdb),
  2. runs the reference: get_model(ci,cs,nh,nb) -> init(db) -> save(dir)        (kmodel.hpp:674,57,173)
     and get_model(dir) -> kmer_to_occ(vector<string>, 8)                         (kmodel.hpp:680,90)
  3. runs the CPU oracle (oracle/kmx_oracle.c) on the same input and REFUSES to write goldens unless
     header/km.bin/rest.bin and the occ vector are identical,
  4. records sha256 of the three model files and of the int32 occ vector in golden.json.
The smallest case is also committed whole (database + model files + query strings + occ vector) under
tests/golden/tiny/.  Only data is stored -- no reference source.
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from kmcex_amd import kmcdb, synth  # noqa: E402

from common import CASES, GENOME_CASES, KMC2_CASES, genome_query_set, query_set, sha_file  # noqa: E402


def main():
    if not O.have_ref():
        sys.exit("oracle/_ref/ref_driver missing: run `make -C oracle ref` where /root/reference exists")
    out = {"generator": "tests/golden/make_golden.py", "seed_k": 1, "seed_c": 2, "cases": {}}
    tmp = tempfile.mkdtemp(prefix="kmx_golden_")
    for name, k, ci, cs, nh, nb, n in CASES:
        km, cnt = synth.make_stream(n, k, ci, cs)
        db = os.path.join(tmp, name)
        kmcdb.write_kmc1(db, km, cnt, k, ci, cs)
        O.ref_build(db, db + ".ref", ci, cs, nh, nb)
        m = O.OracleModel(ci, cs, nh, nb)
        m.build(k, km, cnt)
        m.save(db + ".ora")
        q = query_set(km, k)
        qs = synth.to_strings(q, k)
        r_ref = O.ref_query(db + ".ref", qs, db)
        r_ora = m.query_packed(k, q)
        files = {}
        for f in ("header", "km.bin", "rest.bin"):
            a, b = sha_file(f"{db}.ref/{f}"), sha_file(f"{db}.ora/{f}")
            if a != b:
                sys.exit(f"{name}: oracle {f} differs from the reference")
            files[f] = a
        if not np.array_equal(r_ref, r_ora):
            sys.exit(f"{name}: oracle kmer_to_occ differs from the reference")
        st = m.stats()
        out["cases"][name] = {
            "k": k, "ci": ci, "cs": cs, "nh": nh, "nb": nb, "n_draws": n, "n_kmers": int(len(cnt)),
            "sha256": files, "km_bin_bytes": os.path.getsize(f"{db}.ref/km.bin"),
            "rest_bin_bytes": os.path.getsize(f"{db}.ref/rest.bin"),
            "n_queries": int(len(q)), "occ_sha256": hashlib.sha256(r_ref.astype("<i4").tobytes()).hexdigest(),
            "occ_sum": int(r_ref.astype(np.int64).sum()), "occ_nonzero": int((r_ref != 0).sum()),
            "stats": {"n_km": st.n_km, "n_bf": list(st.n_bf), "attempts": st.attempts,
                      "successes": st.successes, "rest_entries": st.rest_entries},
        }
        print(name, "ok", out["cases"][name]["stats"], flush=True)
        if name == "tiny_k31":
            d = os.path.join(HERE, "tiny")
            shutil.rmtree(d, ignore_errors=True)
            os.makedirs(d)
            for ext in (".kmc_pre", ".kmc_suf"):
                shutil.copy(db + ext, os.path.join(d, "db" + ext))
            for f in ("header", "km.bin", "rest.bin"):
                shutil.copy(f"{db}.ref/{f}", os.path.join(d, f))
            with open(os.path.join(d, "queries.txt"), "w") as f:
                f.write("\n".join(qs) + "\n")
            np.savetxt(os.path.join(d, "occ.txt"), r_ref, fmt="%d")
        m.close()
    out["kmc2_cases"] = {}
    for name, k, ci, cs, nh, nb, n, n_bins in KMC2_CASES:
        km, cnt = synth.make_stream(n, k, ci, cs)
        db = os.path.join(tmp, name)
        order = kmcdb.write_kmc2(db, km, cnt, k, ci, cs, n_bins=n_bins)
        O.ref_build(db, db + ".ref", ci, cs, nh, nb)
        W = synth.words_for_k(k)
        lkm = km.reshape(len(cnt), -1)[order]
        lkm = lkm[:, 0] if W == 1 else lkm
        m = O.OracleModel(ci, cs, nh, nb)
        m.build(k, np.ascontiguousarray(lkm), np.ascontiguousarray(cnt[order]))       # the listing order of the database
        m.save(db + ".ora")
        files = {}
        for f in ("header", "km.bin", "rest.bin"):
            a, b = sha_file(f"{db}.ref/{f}"), sha_file(f"{db}.ora/{f}")
            if a != b:
                sys.exit(f"{name}: oracle {f} differs from the reference")
            files[f] = a
        q = query_set(km, k)
        r_ref = O.ref_query(db + ".ref", synth.to_strings(q, k), db)
        if not np.array_equal(r_ref, m.query_packed(k, q)):
            sys.exit(f"{name}: oracle kmer_to_occ differs from the reference")
        st = m.stats()
        out["kmc2_cases"][name] = {"k": k, "ci": ci, "cs": cs, "nh": nh, "nb": nb, "n_draws": n, "n_bins": n_bins, "sha256": files,
                                   "occ_sha256": hashlib.sha256(r_ref.astype("<i4").tobytes()).hexdigest(),
                                   "order_sha256": hashlib.sha256(order.astype("<i8").tobytes()).hexdigest(),
                                   "stats": {"attempts": st.attempts, "successes": st.successes, "rest_entries": st.rest_entries}}
        print(name, "ok", out["kmc2_cases"][name]["stats"], flush=True)
        m.close()
    out["genome_cases"] = {}
    for name, k, ci, cs, nh, nb, n_bases in GENOME_CASES:
        km, cnt = synth.genome_stream(n_bases, k, ci, cs)
        db = os.path.join(tmp, name)
        kmcdb.write_kmc1(db, km, cnt, k, ci, cs)
        O.ref_build(db, db + ".ref", ci, cs, nh, nb)
        m = O.OracleModel(ci, cs, nh, nb)
        m.build(k, km, cnt)
        m.save(db + ".ora")
        files = {}
        for f in ("header", "km.bin", "rest.bin"):
            a, b = sha_file(f"{db}.ref/{f}"), sha_file(f"{db}.ora/{f}")
            if a != b:
                sys.exit(f"{name}: oracle {f} differs from the reference")
            files[f] = a
        q = genome_query_set(km, k)
        r_ref = O.ref_query(db + ".ref", synth.to_strings(q, k), db)
        if not np.array_equal(r_ref, m.query_packed(k, q)):
            sys.exit(f"{name}: oracle kmer_to_occ differs from the reference")
        out["genome_cases"][name] = {"k": k, "ci": ci, "cs": cs, "nh": nh, "nb": nb, "n_bases": n_bases, "n_kmers": int(len(cnt)),
                                     "sha256": files, "n_queries": int(len(q)),
                                     "occ_sha256": hashlib.sha256(r_ref.astype("<i4").tobytes()).hexdigest(),
                                     "occ_nonzero": int((r_ref != 0).sum())}
        print(name, "ok", len(cnt), "k-mers,", len(q), "queries,", out["genome_cases"][name]["occ_nonzero"], "non-zero", flush=True)
        m.close()
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
