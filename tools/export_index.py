#!/usr/bin/env python3
"""PostgreSQL tables of the FREDDY extension -> one FRDYIDX1 index file (include/freddy_udf.h), the
"Postgres tables -> flat binary -> HBM loader" step of SURVEY 8f-2.  Counterpart of the reference's
index_creation/database_export.py (which goes the other way: arrays -> INSERTs).

Two sources:
  --dsn "dbname=imdb user=postgres ..."   read the tables through psycopg2 (if it is installed)
  --csv-dir DIR                            read <table>.csv files written by psql, one per table:
        \\copy (SELECT id, encode(vector, 'hex') FROM google_vecs_norm ORDER BY id) TO 'google_vecs_norm.csv' CSV
        \\copy (SELECT pos, code, encode(vector, 'hex'), count FROM pq_codebook) TO 'pq_codebook.csv' CSV
        \\copy (SELECT id, encode(vector, 'hex') FROM pq_quantization) TO 'pq_quantization.csv' CSV
        \\copy (SELECT id, encode(vector, 'hex') FROM coarse_quantization) TO 'coarse_quantization.csv' CSV
        \\copy (SELECT pos, code, encode(vector, 'hex'), count FROM residual_codebook) TO 'residual_codebook.csv' CSV
        \\copy (SELECT id, coarse_id, encode(vector, 'hex') FROM fine_quantization) TO 'fine_quantization.csv' CSV
        \\copy (SELECT pos, code, encode(vector, 'hex'), count FROM codebook_ivpq) TO 'codebook_ivpq.csv' CSV
        \\copy (SELECT pos, code, encode(vector, 'hex') FROM coarse_quantization_ivpq) TO 'coarse_quantization_ivpq.csv' CSV
        \\copy (SELECT id, coarse_id, encode(vector, 'hex') FROM fine_quantization_ivpq) TO 'fine_quantization_ivpq.csv' CSV
        \\copy (SELECT coarse_id, coarse_freq FROM stat_fine_quantization_ivpq_coarse_id) TO 'stat.csv' CSV
    (table names are the defaults of freddy--0.0.1.sql:5-20; a missing file skips its table group)
A bytea vector is the raw little-endian array (vec_to_bytea, freddy.c:1790-1826): float32 for vectors and
codebook entries, int16 for PQ codes (index_utils.c:1078-1106).
"""
import argparse
import csv
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "postgres-word2vec_amd")]

QUERIES = {
    "google_vecs_norm": "SELECT id, vector FROM {t} ORDER BY id",
    "pq_codebook": "SELECT pos, code, vector, count FROM {t}",
    "pq_quantization": "SELECT id, vector FROM {t}",
    "coarse_quantization": "SELECT id, vector FROM {t}",
    "residual_codebook": "SELECT pos, code, vector, count FROM {t}",
    "fine_quantization": "SELECT id, coarse_id, vector FROM {t}",
    "codebook_ivpq": "SELECT pos, code, vector, count FROM {t}",
    "coarse_quantization_ivpq": "SELECT pos, code, vector FROM {t}",
    "fine_quantization_ivpq": "SELECT id, coarse_id, vector FROM {t}",
    "stat": "SELECT coarse_id, coarse_freq FROM {t}",
}
VEC_DTYPE = {"pq_quantization": np.int16, "fine_quantization": np.int16, "fine_quantization_ivpq": np.int16}


def rows_from_csv(path):
    with open(path, newline="") as f:
        return [r for r in csv.reader(f) if r]


def rows_from_db(cur, table, name):
    cur.execute(QUERIES[name].format(t=table))
    return [[bytes(c).hex() if isinstance(c, (bytes, memoryview)) else c for c in row] for row in cur.fetchall()]


def vec_column(rows, col, dtype):
    out = [np.frombuffer(bytes.fromhex(r[col][2:] if r[col].startswith("\\x") else r[col]), dtype=dtype) for r in rows]
    if len({len(v) for v in out}) > 1:
        raise SystemExit("vectors of different lengths in one table")
    return np.stack(out) if out else np.zeros((0, 0), dtype)


def int_column(rows, col):
    return np.array([int(r[col]) for r in rows], np.int32)


def collect(get):
    """get(name) -> rows or None; returns the (name, array) list of the file."""
    arrays = []

    def put(name, a):
        arrays.append((name, np.ascontiguousarray(a)))

    r = get("google_vecs_norm")
    if r is not None:
        put("google_vecs_norm.id", int_column(r, 0)); put("google_vecs_norm.vector", vec_column(r, 1, np.float32))
    for cb, q, with_cell in (("pq_codebook", "pq_quantization", False), ("residual_codebook", "fine_quantization", True),
                             ("codebook_ivpq", "fine_quantization_ivpq", True)):
        c, rows = get(cb), get(q)
        if c is None or rows is None:
            continue
        put(cb + ".pos", int_column(c, 0)); put(cb + ".code", int_column(c, 1)); put(cb + ".vector", vec_column(c, 2, np.float32))
        if len(c[0]) > 3:
            put(cb + ".count", int_column(c, 3))
        put(q + ".id", int_column(rows, 0))
        if with_cell:
            put(q + ".coarse_id", int_column(rows, 1))
        put(q + ".vector", vec_column(rows, 2 if with_cell else 1, VEC_DTYPE[q]))
    r = get("coarse_quantization")
    if r is not None:
        put("coarse_quantization.id", int_column(r, 0)); put("coarse_quantization.vector", vec_column(r, 1, np.float32))
    r = get("coarse_quantization_ivpq")
    if r is not None:
        put("coarse_quantization_ivpq.pos", int_column(r, 0)); put("coarse_quantization_ivpq.code", int_column(r, 1))
        put("coarse_quantization_ivpq.vector", vec_column(r, 2, np.float32))
    r = get("stat")
    if r is not None:
        put("stat.coarse_id", int_column(r, 0)); put("stat.coarse_freq", np.array([float(x[1]) for x in r], np.float32))
    return arrays


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--dsn")
    ap.add_argument("--csv-dir")
    ap.add_argument("--out", required=True)
    ap.add_argument("--table", action="append", default=[], metavar="NAME=TABLE", help="non-default table name, e.g. stat=stat_fq_ivpq_coarse_id")
    a = ap.parse_args()
    names = {n: n for n in QUERIES}
    names["stat"] = "stat_fine_quantization_ivpq_coarse_id"
    for kv in a.table:
        k, v = kv.split("=", 1)
        names[k] = v
    if a.dsn:
        try:
            import psycopg2
        except ImportError:
            raise SystemExit("--dsn needs psycopg2; use --csv-dir with psql \\copy dumps instead")
        cur = psycopg2.connect(a.dsn).cursor()

        def get(name):
            try:
                return rows_from_db(cur, names[name], name)
            except Exception:
                cur.connection.rollback()
                return None
    elif a.csv_dir:
        def get(name):
            p = os.path.join(a.csv_dir, name + ".csv")
            return rows_from_csv(p) if os.path.exists(p) else None
    else:
        raise SystemExit("give --dsn or --csv-dir")
    from freddy_amd import udf
    arrays = collect(get)
    if not arrays:
        raise SystemExit("no table found")
    udf.write_index_file(a.out, dict(arrays))
    print(f"wrote {a.out}: " + ", ".join(f"{n}{list(v.shape)}" for n, v in arrays))


if __name__ == "__main__":
    main()
