-- freddy_gpu_watch.sql -- change tracking for the tables the GPU-backed SRFs pin into HBM.
--
-- The reference re-reads its index tables through SPI on EVERY call (freddy.c:69, :239-241, :746-749;
-- ivpq_search_in.c:218-232), so it always sees the current rows.  The GPU hosts pin the tables once per backend;
-- to answer from current data they must notice changes made by ANY backend: insert_batch (freddy.c:1403-1658) appends
-- rows and UPDATEs codebook entries, ordinary DML may do anything, set_*() (freddy--0.0.1.sql:21-132) may point a
-- role at another table (the getters return regclass: a handle remembers the OIDs it was built from).  Plain DML sends no relcache invalidation, so the tables carry statement-level triggers
-- that bump a generation row; pg/freddy_gpu_glue.c reads the rows of a handle's tables before every search
-- (one indexed lookup per table, under the caller's snapshot -- the generation a search sees is the generation of
-- the rows it would have read) and
--   appends only   -> fetches the rows with id > the largest pinned id and appends them in HBM (freddy_gpu_append_rows)
--   codebook UPDATE -> re-reads the (small) codebook table and re-derives its tables (freddy_gpu_update_codebook)
--   anything else  -> unpins and pins again.
-- Without this script the glue falls back to a weaker stamp: table OID, relfilenode, pg_relation_size, max(id)
-- and sum(count) of the codebooks -- it catches insert_batch and TRUNCATE / rewrites, not an in-place UPDATE of a row.
--
--   psql -f pg/freddy_gpu_watch.sql ; SELECT freddy_gpu_watch_all();     -- once per database, after init_function_data

CREATE TABLE IF NOT EXISTS freddy_gpu_generation (
    tab      oid PRIMARY KEY,
    appends  bigint NOT NULL DEFAULT 0,     -- INSERT statements
    rewrites bigint NOT NULL DEFAULT 0      -- UPDATE / DELETE / TRUNCATE statements
);

CREATE OR REPLACE FUNCTION freddy_gpu_bump() RETURNS trigger AS $$
BEGIN
    INSERT INTO freddy_gpu_generation AS g (tab, appends, rewrites)
    VALUES (TG_RELID, (TG_OP = 'INSERT')::int, (TG_OP <> 'INSERT')::int)
    ON CONFLICT (tab) DO UPDATE SET appends = g.appends + EXCLUDED.appends, rewrites = g.rewrites + EXCLUDED.rewrites;
    RETURN NULL;
END
$$ LANGUAGE plpgsql;

CREATE OR REPLACE FUNCTION freddy_gpu_watch(t regclass) RETURNS void AS $$
BEGIN
    EXECUTE format('DROP TRIGGER IF EXISTS freddy_gpu_bump_dml ON %s', t);
    EXECUTE format('DROP TRIGGER IF EXISTS freddy_gpu_bump_truncate ON %s', t);
    EXECUTE format('CREATE TRIGGER freddy_gpu_bump_dml AFTER INSERT OR UPDATE OR DELETE ON %s FOR EACH STATEMENT EXECUTE PROCEDURE freddy_gpu_bump()', t);
    EXECUTE format('CREATE TRIGGER freddy_gpu_bump_truncate AFTER TRUNCATE ON %s FOR EACH STATEMENT EXECUTE PROCEDURE freddy_gpu_bump()', t);
    INSERT INTO freddy_gpu_generation (tab) VALUES (t::oid) ON CONFLICT DO NOTHING;
END
$$ LANGUAGE plpgsql;

-- every table a pinned handle is built from, as the get_*() functions name them now (freddy--0.0.1.sql:134-186);
-- run it again after a set_*() call that introduces a table not watched before
CREATE OR REPLACE FUNCTION freddy_gpu_watch_all() RETURNS void AS $$
DECLARE
    t regclass;
BEGIN
    FOREACH t IN ARRAY ARRAY[get_vecs_name(), get_vecs_name_pq_quantization(), get_vecs_name_codebook(),
                             get_vecs_name_residual_quantization(), get_vecs_name_coarse_quantization(),
                             get_vecs_name_residual_codebook(), get_vecs_name_ivpq_quantization(),
                             get_vecs_name_ivpq_codebook(), get_vecs_name_coarse_quantization_multi(),
                             get_statistics_table()]
    LOOP
        PERFORM freddy_gpu_watch(t);
    END LOOP;
END
$$ LANGUAGE plpgsql;
