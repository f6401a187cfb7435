"""search -> orthology relations without the text round trip (SURVEY.md 8f-1 / BASELINE config 5).

The reference pipeline writes the 16-column .sc file and bin/find_orth.py parses it back.  Here the hit records the
search produced (fixed-width so_hit structs, `Hits.array()`; on several GPUs the records gathered over RCCL) are handed
to the columnar find_orth stage directly; the .sc file is still written when asked for (it is the drop-in artefact), but
nothing reads it.  `orthology_from_search()` is what `bin/find_hit.py ... && bin/find_orth.py -i x.sc` computes."""
import time


def orthology_from_search(fasta_path, sc_path=None, coverage=.5, identity=0., norm='no', sep='|', device=0, **search_kw):
    """self-search of one proteome on the GPU, then IP / OT / CO relations from the hit RECORDS.
    -> (relation lines as bytes, dict of stage wall times in seconds)"""
    from . import find_orth, fsearch
    t = {}
    t0 = time.time()
    data = open(fasta_path, 'rb').read()
    ids = find_orth.fasta_ids(data)
    s = fsearch.Searcher(device=device, **search_kw)
    try:
        s.load_ref_bytes(data)
        s.load_queries_bytes(data)
        t['load'] = time.time() - t0
        t0 = time.time()
        hits = s.search()
        t['search'] = time.time() - t0
        t0 = time.time()
        rec = hits.array()
        if sc_path:
            hits.write(sc_path, 'w')
            t['write_sc'] = time.time() - t0
            t0 = time.time()
        hits.close()
    finally:
        s.close()
    lines = find_orth.relations_from_records(rec, ids, ids, coverage, identity, norm, sep)
    t['find_orth'] = time.time() - t0
    t['rows'] = len(rec)
    return lines, t
