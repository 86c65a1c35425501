"""Shared body of the chain-verification tests: run against the development emulator on CPU (tests/test_cabi_and_host.py) and
against the HIP engine on the GPU box (tests/test_gpu_round4.py) -- same cases, same expectations."""


def run_chain_cases(eng, coracle, cvn, m, n, L, T, keyed, group=0):
    """mp_verify_shuffle_chain: honest chains pass; a bad link gets exactly the per-link verifier's status words; errors that would
    cancel under equal weights are caught; links of one table may name different keys; and a cheating prover whose inner link is made
    under the table's key while its transcript absorbs another key is rejected exactly as it is link by link.
    group > 1 (round 5): the chains of `group` tables share one equation (mp_set_chain_group) -- same verdicts, and a failing equation
    sends the links of ITS tables through the per-link verifier."""
    N, pb = m * n, eng.point_bytes
    g0 = coracle.gen_inputs(cvn, m, n, 50)
    params = g0["params"]
    keys_t = [coracle.gen_inputs(cvn, m, n, 60 + t)["pk"] for t in range(T)] if keyed else [g0["pk"]] * T
    table = eng.table(m, n, params, None if keyed else g0["pk"])
    assert table.chain_group_size(T, L, keyed) == 1       # by size: a handful of tables has nothing to group
    if group > 1:
        assert T % group == 0
        table.set_chain_group(group)
        assert table.chain_group_size(T, L, keyed) == group
    chain = [[coracle.gen_inputs(cvn, m, n, 70 + t)["deck"] for t in range(T)]]     # deck 0 of every table
    proofs = []
    for j in range(L):
        nxt, prf = [], []
        for t in range(T):
            gi = coracle.gen_inputs(cvn, m, n, 100 + 10 * j + t)
            d, p = coracle.shuffle_and_remask(cvn, m, n, params, keys_t[t], chain[j][t], gi["rho"], gi["perm"], gi["prover_seed"])
            nxt.append(d)
            prf.append(p)
        chain.append(nxt)
        proofs.append(prf)
    decks = b"".join(b"".join(row) for row in chain)
    pf = b"".join(b"".join(row) for row in proofs)
    keys = b"".join(keys_t[t] for j in range(L) for t in range(T)) if keyed else None
    eng.profile_enable(True)
    assert table.verify_shuffle_chain(T, L, decks, pf, keys) == [0] * (L * T)
    rep = eng.profile_report()
    table.set_chain_max_links(2)                     # long chains are cut into sub-chains: same verdicts
    try:
        assert table.verify_shuffle_chain(T, L, decks, pf, keys) == [0] * (L * T)
        eng.profile_report()
    finally:
        table.set_chain_max_links(0)
    eng.profile_enable(False)
    assert rep["k_chain_scalars"][0] == 1 and rep["k_bucket_msm"][0] == 1 and "k_var_msm" not in rep and "k_table" not in rep
    # break link 1 of table 0 (swap in another table's proof) and one deck point of the last link of table T-1
    bad = [row[:] for row in proofs]
    bad[1 % L][0] = proofs[1 % L][(0 + 1) % T]
    looked = table.reverified_count()
    st = table.verify_shuffle_chain(T, L, decks, b"".join(b"".join(row) for row in bad), keys)
    # (round 5) the L links of the failing table -- of the `group` tables that share its equation -- were looked at again, nobody else's
    assert table.reverified_count() - looked == L * max(1, group)
    exp = []
    for j in range(L):
        ks = b"".join(keys_t) if keyed else None
        row = (table.verify_shuffle_batch_keys(ks, b"".join(chain[j]), b"".join(chain[j + 1]), b"".join(bad[j])) if keyed else
               table.verify_shuffle_batch(b"".join(chain[j]), b"".join(chain[j + 1]), b"".join(bad[j])))
        exp += row
    assert st == exp and st[(1 % L) * T + 0] > 0 and sum(1 for v in st if v) == 1
    if T > 1:
        # (round 5) the same call in two passes of T - 1 tables and one (mp_set_chain_slice: rows gathered from the link-major arrays)
        table.set_chain_slice(T - 1)
        try:
            assert table.verify_shuffle_chain(T, L, decks, pf, keys) == [0] * (L * T)
            assert table.verify_shuffle_chain(T, L, decks, b"".join(b"".join(row) for row in bad), keys) == exp
            assert table.chain_last_slice() == T - 1
        finally:
            table.set_chain_slice(0)
    # errors that cancel between two links under equal weights are caught: the weights depend on every proof of the chain
    tam = bytearray(decks)
    tam[(1 * T + 0) * N * 2 * pb] ^= 1                 # first byte of deck 1 of table 0: not a curve point any more (or another one)
    st2 = table.verify_shuffle_chain(T, L, bytes(tam), pf, keys)
    assert st2[0 * T + 0] != 0 and st2[1 * T + 0] != 0 and all(v == 0 for i, v in enumerate(st2) if i % T != 0)
    if keyed and L >= 2:
        # the links of a table need not name the same key.  (i) the verifier is told another key for link 1 of table 0 than the one the
        # link was proven under: exactly that link fails, as it does link by link; (ii) link 1 of table 0 really IS proven under
        # another key and the verifier is told so: every link passes, although the chain equation's single key term does not apply
        other = coracle.gen_inputs(cvn, m, n, 99)["pk"]
        klist = [[keys_t[t] for t in range(T)] for j in range(L)]
        klist[1][0] = other
        mixed = b"".join(b"".join(row) for row in klist)
        st3 = table.verify_shuffle_chain(T, L, decks, pf, mixed)
        exp3 = []
        for j in range(L):
            exp3 += table.verify_shuffle_batch_keys(b"".join(klist[j]), b"".join(chain[j]), b"".join(chain[j + 1]), b"".join(proofs[j]))
        assert st3 == exp3 and st3[1 * T + 0] != 0 and sum(1 for v in st3 if v) == 1
        ch2 = [row[:] for row in chain]
        pf2 = [row[:] for row in proofs]
        for j in range(1, L):                          # table 0 from link 1 on: link 1 under `other`, the rest under the table's key again
            gi = coracle.gen_inputs(cvn, m, n, 123 + j)
            ch2[j + 1][0], pf2[j][0] = coracle.shuffle_and_remask(cvn, m, n, params, other if j == 1 else keys_t[0], ch2[j][0], gi["rho"],
                                                                   gi["perm"], gi["prover_seed"])
        st4 = table.verify_shuffle_chain(T, L, b"".join(b"".join(r) for r in ch2), b"".join(b"".join(r) for r in pf2), mixed)
        assert st4 == [0] * (L * T)
        # (iii) a cheating prover: link 1 of table 0 is made under the table's key but its transcript absorbs `other`, and the
        # verifier is told `other` for that link.  Link by link the multi-exponentiation check fails (the algebra runs under
        # `other`); a chain equation that took link 0's key for every link would accept it.
        import mp_oracle as po
        cv = po.CURVES[cvn]
        with po.curve_ctx(cv):
            w = po.point_bytes()
            pts = [po.pt_from_wire(params[i:i + w]) for i in range(0, len(params), w)]
            pp = po.Params(cv, m, n, pts[0], pts[1:1 + n], pts[1 + n], pts[2 + n])
            gi = coracle.gen_inputs(cvn, m, n, 777)
            rho = [int.from_bytes(gi["rho"][i:i + 32], "little") for i in range(0, 32 * N, 32)]
            honest_statement = po.statement_bytes
            po.statement_bytes = lambda pp_, pk_, d_, s_: honest_statement(pp_, po.pt_from_wire(other), d_, s_)
            try:
                sh, prf = po.shuffle_and_remask(pp, po.pt_from_wire(keys_t[0]), po.deck_from_bytes(chain[1][0]), rho, list(gi["perm"]), gi["prover_seed"])
            finally:
                po.statement_bytes = honest_statement
            ch3 = [row[:] for row in chain]
            pf3 = [row[:] for row in proofs]
            ch3[2][0], pf3[1][0] = po.deck_to_bytes(sh), po.proof_to_bytes(prf)
        for j in range(2, L):
            gj = coracle.gen_inputs(cvn, m, n, 800 + j)
            ch3[j + 1][0], pf3[j][0] = coracle.shuffle_and_remask(cvn, m, n, params, keys_t[0], ch3[j][0], gj["rho"], gj["perm"], gj["prover_seed"])
        st5 = table.verify_shuffle_chain(T, L, b"".join(b"".join(r) for r in ch3), b"".join(b"".join(r) for r in pf3), mixed)
        exp5 = []
        for j in range(L):
            exp5 += table.verify_shuffle_batch_keys(b"".join(klist[j]), b"".join(ch3[j]), b"".join(ch3[j + 1]), b"".join(pf3[j]))
        assert st5 == exp5 and st5[1 * T + 0] != 0 and sum(1 for v in st5 if v) == 1
