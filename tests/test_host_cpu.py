"""Host logic on CPU (drivers, batching, finishing, I/O trimming, CLI rows) against vectors the
reference produced.  Device work is answered by tests/fake_engine.py (oracle-backed, test only)."""
import os

import numpy as np
import pytest

from conftest import load_golden
from fake_engine import FakeEngine
from vapor_amd import cli, pipeline, seqio, synth


@pytest.fixture()
def fake(oracle):
    e = FakeEngine(oracle)
    pipeline.set_engine(e)
    yield e
    pipeline.set_engine(None)
    seqio.set_backend(None)


IO = load_golden("io.json.gz")


@pytest.mark.parametrize("case", IO["cases"], ids=lambda c: c["fn"])
def test_io_helpers(case):
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(IO["world"])))
    fn = getattr(seqio, case["fn"])
    exp = case["out"]
    try:
        if "error" in exp:
            with pytest.raises(Exception) as ei:
                fn(*case["args"])
            assert type(ei.value).__name__ == exp["error"]
        else:
            assert fn(*case["args"]) == exp["ok"]
    finally:
        seqio.set_backend(None)


@pytest.mark.parametrize("case", load_golden("window.json.gz")["cases"], ids=lambda c: c["name"])
def test_window_size_refine(fake, case, monkeypatch):
    """window_size_refine (SF:2030-2046) against the reference's answers.  The windows that pass through its X-means branch
    (unseeded sklearn / scipy draws, SF:856-887) were run by the reference under several numpy seeds when the fixture was
    made: where every seed gave the same answer (`xmeans_seed_independent`: BIC keeps one cluster - all tandem-duplication alt
    windows are of that kind) the answer is pinned and the product must give it whatever its own seed; where the reference
    itself raises on this SciPy (`scipy.std`, SF:878, is gone: every split into more than one cluster ends there) the product
    goes on as the reference did on the SciPy it was written for - a stated deviation, recorded here."""
    from vapor_amd import simple_function as SF
    if case["xmeans_calls"] > 0 and not case["xmeans_seed_independent"]:
        assert case["xmeans_reference_raises"] == ["AttributeError"] and case.get("error") == "AttributeError"
        answers = []
        for seed in ("7", "8"):
            monkeypatch.setenv("VAPOR_QC_SEED", seed)
            got = SF.window_size_refine(case["seq"])
            assert got[0] in (10, 20, 30, 40) and 0.0 < float(got[1][0]) <= 1.0 and len(got[1][1]) >= 1
            # the integer part of the check is the reference's (its qc_trace): the diagonal share of the first self plot
            if got[0] == 10:
                assert float(got[1][0]) == case["qc_trace"][0][1] / case["qc_trace"][0][0]
            answers.append(got)
        monkeypatch.setenv("VAPOR_QC_SEED", "7")
        assert SF.window_size_refine(case["seq"]) == answers[0]          # (the same seed, the same answer)
        return
    if "error" in case:
        with pytest.raises(Exception):
            SF.window_size_refine(case["seq"])
        return
    for seed in ((None,) if case["xmeans_calls"] == 0 else ("7", "8", None)):
        if seed is None:
            monkeypatch.delenv("VAPOR_QC_SEED", raising=False)
        else:
            monkeypatch.setenv("VAPOR_QC_SEED", seed)
        got = SF.window_size_refine(case["seq"])
        assert got[0] == case["window_size"]
        if case["window_size"] != "Error":
            assert float(got[1][0]) == float(case["qc"][0]) and [float(v) for v in got[1][1]] == [float(v) for v in case["qc"][1]]


def test_window_fixture_pins_the_xmeans_outcomes_that_are_pinnable():
    """at most three of the windows that meet the X-means branch stay unpinned (the ones where the
    reference itself raises), and every tandem-duplication alt window is pinned."""
    cases = load_golden("window.json.gz")["cases"]
    xm = [c for c in cases if c["xmeans_calls"] > 0]
    assert len(xm) >= 18
    assert sorted(c["name"] for c in xm if not c["xmeans_seed_independent"]) == ["N_100_ok", "inverted_repeat", "tandem_500x2"]
    assert all(c["xmeans_seed_independent"] for c in xm if c["name"].startswith("tandup_alt_"))


@pytest.mark.parametrize("case", load_golden("scorers.json.gz")["cases"], ids=lambda c: c["name"])
def test_scorer_functions(fake, case):
    from vapor_amd import simple_function as SF
    x = [case["read"], case["miss"], case["name"]]
    for key, fn in (("s1", SF.calcu_vapor_single_read_score_abs_dis_m1b),
                    ("s2", SF.calcu_vapor_single_read_score_within_10Perc_m1b),
                    ("s3", SF.calcu_vapor_single_read_score_directed_dis_m1b_redefine_diagnal)):
        if "error" in case[key]:
            with pytest.raises(KeyError):
                fn(case["ref"], case["alt"], x, case["k"])
        else:
            assert [float(v) for v in fn(case["ref"], case["alt"], x, case["k"])] == [float(v) for v in case[key]["ok"]]


LOCUS = load_golden("locus_bed.json.gz")["cases"] + load_golden("locus_long.json.gz")["cases"]     # (+ spans of 20-99 kb: the junction-window branches)


@pytest.mark.parametrize("case", LOCUS, ids=lambda c: c["name"])
def test_bed_cli_rows(fake, case, tmp_path):
    """`vapor bed` end to end on an in-memory world: the .vapor table must equal the reference's,
    byte for byte, and the loci must have gone to the device in shared batches."""
    world = synth.world_from_json(case["world"])
    seqio.set_backend(seqio.MemorySamtools(world))
    bed = tmp_path / "in.bed"
    bed.write_text(case["bed"])
    out = tmp_path / "out.vapor"
    ref_failed = [p for p in case["per_locus"] if "error" in p["scores"]]
    args = ["bed", "--sv-input", str(bed), "--reference", "ref.fa", "--pacbio-input", "x.bam",
            "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]
    if ref_failed:
        # The reference died inside its X-means branch for these loci (`scipy.std`, SF:878, is gone from current SciPy; the
        # fixture's generator catches the exception per locus and goes on).  The product clusters as the reference did on the
        # SciPy it was written for and scores the locus: its table is the reference's plus one row per such locus.
        assert all(p["scores"]["error"] == "AttributeError" and p.get("xmeans_calls", 1) > 0 for p in ref_failed)
        assert cli.main(args) == 0
        got = out.read_text().splitlines()
        exp = case["vapor_text"].splitlines()
        failed = {tuple(p["key"].split(":")) for p in ref_failed}
        assert [ln for ln in got if tuple(ln.split("\t")[:4]) not in failed] == exp
        assert len(got) == len(exp) + len(failed)
        return
    for seed in (("7", "8") if case.get("xmeans_seed_independent") else (None,)):
        # (loci whose windows meet the reference's unseeded X-means: the fixture's rows were the same under every seed the
        # reference was run with, so the product must give them whatever its own seed - every TANDUP row is of that kind)
        if seed is not None:
            os.environ["VAPOR_QC_SEED"] = seed
        try:
            assert cli.main(args) == 0
        finally:
            os.environ.pop("VAPOR_QC_SEED", None)
        assert out.read_text() == case["vapor_text"]
    n_loci = len(case["per_locus"])
    assert len(fake.batches) < 4 * n_loci or n_loci < 3     # batched, not one plan per read


def test_tandup_rows_are_pinned_end_to_end():
    """tandem-duplication loci whose alt window enters the X-means band have pinned final rows - the
    reference gives the same table under every seed it was run with (tests/golden/locus_bed.json.gz)."""
    cases = {c["name"]: c for c in LOCUS}
    band = cases["bed_tandup_band"]
    assert band["xmeans_seed_independent"] is True
    assert all(p["xmeans_calls"] >= 1 and len(p["scores"]["ok"]) >= 4 for p in band["per_locus"])
    assert band["vapor_text"].count("TANDUP") == 6 and "\tNA" not in band["vapor_text"]


def test_driver_functions_match_reference_per_locus(fake):
    """The reference-named one-locus drivers return the reference's score lists."""
    from vapor_amd import simple_function as SF
    case = LOCUS[0]
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
    for p in case["per_locus"]:
        row = p["bed_row"]
        if row[-1] == "/a":
            got = SF.vapor_simple_del_Vapor(3, 1, "x.bam", "ref.fa", row[:3], "f.png")
        elif row[-1] == "a/a^":
            got = SF.vapor_simple_inv_Vapor(3, 1, "x.bam", "ref.fa", row[:3], "f.png")
        elif row[-1] == "a/aa":
            got = SF.vapor_simple_tandup_Vapor(3, 1, "x.bam", "ref.fa", row[:3], "f.png")
        else:
            got = SF.vapor_simple_ins_Vapor(3, 1, "x.bam", "ref.fa", "%s_%d" % (row[0], row[1]), row[4], "f.png", "+")
        assert [float(v) for v in got] == [float(v) for v in p["scores"]["ok"]]


def test_partition_is_balanced_and_complete():
    from vapor_amd import dist
    costs = [5, 1, 1, 9, 2, 2, 7, 3]
    parts = dist.partition(costs, 3)
    assert sorted(sum(parts, [])) == list(range(len(costs)))
    loads = [sum(costs[t] for t in p) for p in parts]
    assert max(loads) - min(loads) <= max(costs)
    flat, extra = dist.pack_records({0: [0.5, -1.25], 2: [], 3: KeyError("x"), 5: list(range(40))}, [0, 2, 3, 5])
    assert flat.tolist()[:6] == [2.0, 0.5, -1.25, 0.0, -1.0, 40.0] and list(extra) == [3]
    back = [None] * 6
    dist.unpack_records(flat, [0, 2, 3, 5], extra, back)
    assert back[0] == [0.5, -1.25] and back[2] == [] and isinstance(back[3], KeyError) and back[5] == [float(v) for v in range(40)]
    assert dist.gather_results({0: [1.0], 1: []}, 2) == [[1.0], []]


VCF = load_golden("locus_vcf.json.gz")["cases"]
_VCF_WORLDS = {c["name"]: c["world"] for c in VCF if c["world"] is not None}


def _vcf_world(case):
    return synth.world_from_json(_VCF_WORLDS[case["world_of"]])


def test_vcf_parser_drops_the_duplicates_the_reference_drops(tmp_path):
    """cli.vcf_list_readin answers the reference's `x not in out[T]` (vapor_vali/vapor:127-202: a scan of the bucket's list
    per record) from sets beside the lists: the buckets and the record keys of a VCF with repeated records of every type
    equal those of the scans written out - including the repeats the reference never drops (an insertion's probe has three
    fields, its entries four)."""
    from vapor_amd import simple_function as SF
    rows = []
    rng = np.random.default_rng(8)
    for t in range(400):
        c = "chr%d" % rng.integers(1, 4)
        a = int(rng.integers(1, 40)) * 100
        b = a + int(rng.integers(1, 5)) * 50
        kind = t % 7
        if kind == 0:
            info = "SVTYPE=DEL;END=%d" % b
        elif kind == 1:
            info = "SVTYPE=INV;END=%d" % b
        elif kind == 2:
            info = "SVTYPE=INS;END=%d;SVLEN=%d" % (a + 1, int(rng.integers(1, 3)) * 100)
        elif kind == 3:
            info = "SVTYPE=DUP;END=%d" % b
        elif kind == 4:
            info = "SVTYPE=DISDUP;END=%d;insert_point=%s:%d" % (b, c, b + 1000)
        elif kind == 5:
            info = "SVTYPE=DEL_INV;END=%d;del=%s:%d-%d;inv=%s:%d-%d" % (b + 200, c, a, b, c, b, b + 200)
        else:
            info = "SVTYPE=CX;END=%d;Other=ab/ab_b/b^_%s:%d:%d:%d" % (b + 100, c, a, b, b + 100)
        rows.append("%s\t%d\tid%d\tN\t<X>\t.\tPASS\t%s" % (c, a, t, info))
    vcf = tmp_path / "dups.vcf"
    vcf.write_text("\n".join(rows) + "\n")
    got, got_keys = cli.vcf_list_readin(str(vcf))
    # the scans, written out
    exp, exp_keys = {}, {}
    for rec, line in enumerate(rows):
        pin = line.split()
        t = SF.svtype_extract(pin)
        pos = SF.chr_start_end_extract(pin)
        if t == "DEL" or t == "INV":
            exp.setdefault(t, [])
            if pos not in exp[t]:
                exp[t].append(pos); exp_keys[rec] = ":".join([str(i) for i in pos] + [t])
        elif t == "INS":
            n = int(SF.sv_len_extract(pin))
            exp.setdefault("INS", [])
            if n > 0 and pos not in exp["INS"]:
                exp["INS"].append(pos[:2] + [n, SF.sv_seq_extract(pin)]); exp_keys[rec] = ":".join([str(i) for i in pos[:2] + [n]] + ["INS"])
        elif t == "DUP":
            exp.setdefault("TANDUP", [])
            if pos not in exp["TANDUP"]:
                exp["TANDUP"].append(pos); exp_keys[rec] = ":".join([str(i) for i in pos] + ["TANDUP"])
        elif t == "DISDUP":
            ip = SF.sv_insert_point_define(pin)
            exp.setdefault("DISDUP", [])
            if pos not in exp["DISDUP"]:
                exp["DISDUP"].append(pos + ip); exp_keys[rec] = ":".join([str(i) for i in pos + ip] + ["DISDUP"])
        elif t == "DEL_INV":
            info = cli.del_inv_interprete(pin)
            exp.setdefault("DEL_INV", [])
            if not info == "error" and info not in exp["DEL_INV"]:
                exp["DEL_INV"].append(info); exp_keys[rec] = ":".join(["_".join([str(i) for i in j]) for j in info] + ["DEL_INV"])
        else:
            o = [i for i in pin[7].split(";") if i[:6] == "Other="][0].split("=")[1].split("_")
            item = ["_".join(i.split("/")) for i in o[:2]] + o[2].split(":")
            exp.setdefault("Other", [])
            if item not in exp["Other"]:
                exp["Other"].append(item); exp_keys[rec] = ":".join([str(i) for i in item + ["CANNOT_CLASSIFY"]])
    assert got == exp and got_keys == exp_keys
    assert len(exp["DEL"]) < 58 and len(exp["INS"]) == 57            # deletions repeat and are dropped; insertions never are


@pytest.mark.parametrize("case", [c for c in VCF if not c["header"]], ids=lambda c: c["name"])
def test_vcf_records_and_table(fake, case, tmp_path):
    """`vapor vcf` on header-less input (where the reference's record numbering is consistent):
    per-record scores, the 6-column table and the INFO-annotated VCF equal the reference's."""
    from vapor_amd import drivers
    world = _vcf_world(case)
    seqio.set_backend(seqio.MemorySamtools(world))
    vcf = tmp_path / "in.vcf"
    vcf.write_text(case["vcf"])
    vcf_list, rec_hash = cli.vcf_list_readin(str(vcf))
    clean = all("ok" in p["scores"] for p in case["per_record"])
    # record by record through the one-locus drivers
    fns = {"DEL": drivers.vapor_simple_del, "INV": drivers.vapor_simple_inv, "DISDUP": drivers.vapor_simple_disdup,
           "DEL_INV": drivers.vapor_del_inv, "DUP_INV": drivers.vapor_dup_inv}
    it = iter(case["per_record"])
    for x in list(vcf_list.keys()):
        for y in vcf_list[x]:
            if x not in ("DEL", "INV", "INS", "DISDUP", "DEL_INV", "DUP_INV"):
                continue
            p = next(it)
            assert p["type"] == x
            if x in ("DEL", "INV") and y[2] - y[1] < 50:
                continue
            if x == "INS":
                gen = drivers.vapor_simple_ins(3, 1, "x.bam", "ref.fa", "_".join(str(i) for i in y[:2]),
                                               y[-1] if len(y) == 4 else "X" * y[2], "f.png", "+")
            else:
                gen = fns[x](3, 1, "x.bam", "ref.fa", y, "f.png")
            if "error" in p["scores"]:
                with pytest.raises(Exception) as ei:
                    pipeline.run_sync(gen)
                assert type(ei.value).__name__ == p["scores"]["error"]
            else:
                got = pipeline.run_sync(gen)
                assert [float(v) for v in got] == [float(v) for v in p["scores"]["ok"]], (x, y)
    if not clean:
        return
    rc = cli.main(["vcf", "--sv-input", str(vcf), "--reference", "ref.fa", "--pacbio-input", "x.bam",
                   "--output-path", str(tmp_path / "figs"), "--output-file", "unused", "--no-figures"])
    assert rc == 0
    assert (tmp_path / "in.vcf.vapor").read_text() == case["final"]


@pytest.mark.parametrize("name", ["vcf_simple", "vcf_tiny_span"])
def test_vcf_with_header_annotates_the_right_records(fake, name, tmp_path):
    """With header lines the reference mis-indexes records (KeyError, see vcf_vapor_modify's
    docstring); here the same records get the same annotation as in the header-less run."""
    hdr = [c for c in VCF if c["name"] == name + "_hdr"][0]
    nohdr = [c for c in VCF if c["name"] == name + "_nohdr"][0]
    assert hdr["final_status"] != "ok"
    seqio.set_backend(seqio.MemorySamtools(_vcf_world(hdr)))
    vcf = tmp_path / "in.vcf"
    vcf.write_text(hdr["vcf"])
    assert cli.main(["vcf", "--sv-input", str(vcf), "--reference", "ref.fa", "--pacbio-input", "x.bam",
                     "--output-path", str(tmp_path / "figs"), "--output-file", "unused", "--no-figures"]) == 0
    got = (tmp_path / "in.vcf.vapor").read_text().splitlines()
    recs = [l for l in got if not l.startswith("#")]
    assert recs == [l for l in nohdr["final"].splitlines() if l and not l.startswith("#")]
    assert sum(1 for l in got if l.startswith("##INFO=<ID=VaPoR_")) == 4
    assert got[[i for i, l in enumerate(got) if l.startswith("#CHROM")][0] - 1].startswith("##source")


OTHER = load_golden("locus_other.json.gz")


@pytest.mark.parametrize("case", OTHER["cases"], ids=lambda c: c["name"])
def test_cannot_classify_driver(fake, case):
    """vapor_CANNOT_CLASSIFY_VapoR on letter structures (whole-region scoring and the junction fallback)."""
    from vapor_amd import simple_function as SF
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(OTHER["world"])))
    got = SF.vapor_CANNOT_CLASSIFY_VapoR(3, 1, "x.bam", "ref.fa", list(case["sv_info"]), "f.png")
    assert [float(v) for v in got] == [float(v) for v in case["scores"]["ok"]]


def test_svelter_mode_rows(fake, tmp_path):
    world = synth.world_from_json(OTHER["world"])
    seqio.set_backend(seqio.MemorySamtools(world))
    sv = tmp_path / "calls.svelter"
    rows = ["chr\tstart\tend\tbp_info\tref\talt\tscore"]
    for c in OTHER["cases"][:4]:
        i = c["sv_info"]
        rows.append("\t".join([i[2], i[3], i[5], ":".join(i[2:]), i[0].replace("_", "/"), i[1].replace("_", "/"), "0"]))
    sv.write_text("\n".join(rows) + "\n")
    out = tmp_path / "out.txt"
    assert cli.main(["svelter", "--sv-input", str(sv), "--reference", "ref.fa", "--pacbio-input", "x.bam",
                     "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]) == 0
    lines = out.read_text().splitlines()
    assert len(lines) == 4
    from vapor_amd import finish, simple_function as SF
    for line, c in zip(lines, OTHER["cases"][:4]):
        exp = SF.format_output_row(finish.result_organize_ins(["." + "_".join(c["sv_info"][2:]), c["scores"]["ok"]]))
        assert line == exp


CFG1 = load_golden("config1_bed.json.gz")


def test_config1_vapor_test_bed_plumbing(fake, tmp_path):
    """BASELINE.json configs[0]: the 19 loci of the reference's vapor_test.bed (re-expressed as the 5-column
    BED today's parser wants) on a stand-in genome: `vapor bed` writes the table the reference writes."""
    world = synth.make_world_from_bed(CFG1["bed_rows"], seed=CFG1["seed"])
    assert synth.bed_text(world) == CFG1["bed"]
    seqio.set_backend(seqio.MemorySamtools(world))
    bed = tmp_path / "vapor_test.bed"
    bed.write_text(CFG1["bed"])
    out = tmp_path / "vapor_test.bed.vapor"
    assert cli.main(["bed", "--sv-input", str(bed), "--reference", "hg19.fa", "--pacbio-input", "x.bam",
                     "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]) == 0
    assert out.read_text() == CFG1["cases"][0]["vapor_text"]


def _run_melt(tmp_path):
    d = load_golden("melt_ins.json.gz")
    world = synth.world_from_json(d["world"])
    world.contigs.update(d["fasta"])
    seqio.set_backend(seqio.MemorySamtools(world))
    prefix = tmp_path / "S1.melt.sites"
    (tmp_path / "S1.melt.sites.vcf").write_text(d["vcf"])
    rc = cli.main(["ins", "--sv-input", str(prefix), "--reference", "ref.fa", "--pacbio-input", "x.bam",
                   "--output-path", str(tmp_path / "figs"), "--output-file", "unused", "--no-figures"])
    return rc, (tmp_path / "S1.melt.sites.vapor").read_text(), d["cases"][0]["vapor_text"]


def test_melt_ins_mode(fake, tmp_path):
    """`vapor ins`: MELT sites + assembled sequences (polarity, N -> X, missing sequence -> X * SVLEN)."""
    rc, got, exp = _run_melt(tmp_path)
    assert rc == 0 and got == exp


def test_chromos_readin_keeps_the_names_until_the_index_changes(tmp_path):
    """seqio.chromos_readin (SF:356-363; the reference reads the .fai again per unclassified record): the names as a list,
    `in` as the list would answer, read once per index file and backend - and again when the file changes or, for an
    in-memory world, when its contig table does."""
    import time
    fa = tmp_path / "ref.fa"
    fai = tmp_path / "ref.fa.fai"
    fai.write_text("chr1\t1000\t6\t60\t61\nchr2 extra\t500\t1100\t60\t61\n\n")
    calls = []

    class Be(seqio.InProcessBam):
        def fai_lines(self, ref):
            calls.append(ref)
            return super().fai_lines(ref)
    seqio.set_backend(Be())
    try:
        a = seqio.chromos_readin(str(fa))
        assert a == ["chr1", "chr2"] and isinstance(a, list) and "chr2" in a and "chr3" not in a and ["x"] not in a and 7 not in a
        assert seqio.chromos_readin(str(fa)) is a and len(calls) == 1
        time.sleep(0.01)
        fai.write_text("chr1\t1000\t6\t60\t61\nchr2\t500\t1100\t60\t61\nchr3\t9\t1700\t60\t61\n")
        assert seqio.chromos_readin(str(fa)) == ["chr1", "chr2", "chr3"] and len(calls) == 2
        w = synth.make_world(seed=5, n_loci=2, svtypes=("DEL",), span_range=(100, 300), read_len=1200, n_reads=3)
        seqio.set_backend(seqio.MemorySamtools(w))
        names = seqio.chromos_readin("ref.fa")
        assert names == list(w.contigs) and seqio.chromos_readin("ref.fa") is names
        w.contigs["later"] = "ACGT"
        assert seqio.chromos_readin("ref.fa") == list(w.contigs) and "later" in seqio.chromos_readin("ref.fa")
    finally:
        seqio.set_backend(None)


def test_fai_fasta_reader_matches_faidx_semantics(tmp_path):
    """In-process .fai reader: 1-based inclusive windows, clipping, and the lines ref_seq_readin parses."""
    rng = np.random.default_rng(3)
    contigs = {"chrA": synth.random_dna(rng, 1234), "chrB_x": synth.random_dna(rng, 61), "chrC": synth.random_dna(rng, 60)}
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as f, open(str(fa) + ".fai", "w") as fi:
        off = 0
        for name, s in contigs.items():
            hdr = ">%s some description\n" % name
            f.write(hdr)
            off += len(hdr)
            fi.write("%s\t%d\t%d\t60\t61\n" % (name, len(s), off))
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
                off += len(s[i:i + 60]) + 1
    r = seqio.FaiFasta(str(fa))
    for name, s in contigs.items():
        for a, b in ((1, 10), (55, 130), (1, len(s)), (len(s) - 5, len(s) + 50), (60, 61), (61, 61), (-5, 3)):
            assert r.fetch(name, a, b) == s[max(a, 1) - 1:min(b, len(s))], (name, a, b)
    lines = r.lines("chrA:100-400")
    assert lines[0] == ">chrA:100-400" and "".join(lines[1:]) == contigs["chrA"][99:400] and all(len(l) <= 60 for l in lines[1:])

    class Be:                                   # a text-only backend: ref_seq_readin parses the faidx lines
        def faidx_lines(self, ref, region):
            return r.lines(region)
    seqio.set_backend(Be())
    try:
        assert seqio.ref_seq_readin(str(fa), "chrA", 100, 400) == contigs["chrA"][99:400]
        assert seqio.ref_seq_readin(str(fa), "chrB_x", 1, 61, "TRUE") == seqio.reverse(seqio.complementary(contigs["chrB_x"]))
    finally:
        seqio.set_backend(None)


def test_inprocess_bam_reader_equals_memory_backend(tmp_path):
    """BAM + BAI written by vapor_amd.bamio from a synthetic world, read back in-process: region queries
    return the records the in-memory backend returns, and the read trimming built on them is identical."""
    from vapor_amd import bamio
    w = synth.make_world(seed=5, n_loci=3, svtypes=("DEL", "INV", "TANDUP"), span_range=(200, 900), read_len=2600, n_reads=30)
    # put all loci on ONE contig 40 kb apart so that bins and the linear index are exercised
    contig = ""
    recs = []
    shift = {}
    for k, (name, seq) in enumerate(w.contigs.items()):
        shift[name] = len(contig)
        contig += seq + synth.random_dna(np.random.default_rng(k), 40000)
    for name, rs in w.reads.items():
        for r in rs:
            recs.append((r.qname, 0, shift[name] + r.pos - 1, r.cigar, r.seq))
    bam = str(tmp_path / "x.bam")
    bamio.write_bam(bam, [("chrU", len(contig))], recs, block_size=4096)
    big = synth.SynthWorld()
    big.contigs["chrU"] = contig
    big.reads["chrU"] = sorted([synth.SamRecord(q, "chrU", p + 1, c, s, sum(int(n) for n, o in
                                __import__("re").findall(r"(\d+)([MD=XN])", c))) for q, _t, p, c, s in recs], key=lambda r: r.pos)
    mem = seqio.MemorySamtools(big)
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as f, open(str(fa) + ".fai", "w") as fi:
        f.write(">chrU\n")
        fi.write("chrU\t%d\t6\t60\t61\n" % len(contig))
        for i in range(0, len(contig), 60):
            f.write(contig[i:i + 60] + "\n")
    inproc = seqio.InProcessBam()
    for l in w.loci:
        s0 = shift[l.chrom]
        for a, b in ((s0 + l.start - 500, s0 + l.start + 500), (s0 + l.start - 500, s0 + l.end + 500), (s0 + 1, s0 + 50)):
            exp = [ln.split("\t") for ln in mem.view_lines("x.bam", "chrU:%d-%d" % (a, b))]
            got = [ln.split("\t") for ln in inproc.view_lines(bam, "chrU:%d-%d" % (a, b))]
            key = lambda t: (int(t[3]), t[0])
            assert sorted([(t[0], t[3], t[5], t[9]) for t in got], key=lambda t: (int(t[1]), t[0])) == \
                sorted([(t[0], t[3], t[5], t[9]) for t in exp], key=lambda t: (int(t[1]), t[0]))
        seqio.set_backend(mem)
        e1 = seqio.simple_chop_pacbio_read_simple_short("x.bam", ["chrU", s0 + l.start, s0 + l.end], 500)
        r1 = seqio.ref_seq_readin(str(fa), "chrU", s0 + l.start - 500, s0 + l.end + 500)
        seqio.set_backend(inproc)
        e2 = seqio.simple_chop_pacbio_read_simple_short(bam, ["chrU", s0 + l.start, s0 + l.end], 500)
        r2 = seqio.ref_seq_readin(str(fa), "chrU", s0 + l.start - 500, s0 + l.end + 500)
        seqio.set_backend(None)
        assert sorted(e1) == sorted(e2) and len(e1) > 3
        assert r1 == r2


def test_workflow_chooses_ranks_per_gpu_from_input_and_cores(tmp_path):
    """The launcher's default for --ranks-per-gpu: one rank per four cores of the host's quota, four at most, 3 000
    records per rank at least - so small inputs still run in the calling process."""
    from vapor_amd import workflow
    f = workflow.auto_ranks_per_gpu
    assert f(19, 1, 16) == 1 and f(2999, 1, 64) == 1 and f(6000, 1, 16) == 2 and f(12000, 1, 16) == 4 and f(10 ** 6, 1, 16) == 4
    assert f(50000, 1, 8) == 2 and f(50000, 1, 3) == 1 and f(50000, 8, 128) == 2 and f(10 ** 6, 8, 128) == 4 and f(10 ** 6, 8, 16) == 1
    assert f(0, 1, 16) == 1 and f(10 ** 6, 0, 16) == 4
    bed = tmp_path / "x.bed"
    bed.write_text("#chr\tstart\n" + "".join("c\t%d\t%d\tDEL\n" % (i, i + 9) for i in range(7)) + "\n\n")
    assert workflow._count_records(str(bed)) == 7 and workflow._count_records(str(tmp_path / "none.bed")) == 0


def test_workflow_launcher_reports_a_failed_rank(tmp_path):
    """The launcher's own ranks (several per GPU): a run whose ranks end badly - here the input does not exist, which every
    rank finds out before it touches a device - returns their exit code, leaves no rank behind and removes its directory."""
    import glob
    import tempfile
    from vapor_amd import workflow
    before = set(glob.glob(os.path.join(tempfile.gettempdir(), "vapor_ranks_*")))
    rc = workflow.main(["--ranks-per-gpu", "2", "--prefix", str(tmp_path / "s"), "bed", "--sv-input", str(tmp_path / "missing.bed"),
                        "--reference", "ref.fa", "--pacbio-input", "x.bam", "--output-path", str(tmp_path / "figs"),
                        "--output-file", str(tmp_path / "out.vapor"), "--no-figures"])
    assert rc != 0
    assert set(glob.glob(os.path.join(tempfile.gettempdir(), "vapor_ranks_*"))) == before
    assert not (tmp_path / "s.bed.gz").exists()


def test_workflow_sorted_bgzipped_indexed_table(fake, tmp_path):
    """§8f-4: the node launcher's gather side - the CLI's table, version-sorted, block-gzipped and tabix-indexed
    (what the reference's WDL does with sort -V | bgzip | tabix -p bed)."""
    import gzip
    from vapor_amd import workflow
    assert sorted(["chr10", "chr2", "chrX", "chr1", "chr2_alt"], key=workflow.version_key) == ["chr1", "chr2", "chr2_alt", "chr10", "chrX"]
    world = synth.make_world_from_bed(CFG1["bed_rows"], seed=CFG1["seed"])
    seqio.set_backend(seqio.MemorySamtools(world))
    bed = tmp_path / "vapor_test.bed"
    bed.write_text(CFG1["bed"])
    out = tmp_path / "s1.vapor"
    assert workflow.main(["--prefix", str(tmp_path / "s1"), "bed", "--sv-input", str(bed), "--reference", "hg19.fa",
                          "--pacbio-input", "x.bam", "--output-path", str(tmp_path / "figs"), "--output-file", str(out),
                          "--no-figures"]) == 0
    table = out.read_text().splitlines()
    assert out.read_text() == CFG1["cases"][0]["vapor_text"]
    gz = str(tmp_path / "s1.bed.gz")
    rows = gzip.open(gz, "rt").read().splitlines()
    assert rows == workflow.sort_rows(table[1:]) and len(rows) == len(table) - 1
    assert open(gz, "rb").read().endswith(workflow._BGZF_EOF)
    # every row is found through the index by its own interval, and only overlapping rows come back
    for ln in rows:
        f = ln.split("\t")
        got = workflow.tabix_query(gz, f[0], int(f[1]) + 1, max(int(f[2]), int(f[1]) + 1))
        assert ln in got
        for g in got:
            h = g.split("\t")
            assert h[0] == f[0] and int(h[1]) < max(int(f[2]), int(f[1]) + 1) and max(int(h[2]), int(h[1]) + 1) > int(f[1])
    assert workflow.tabix_query(gz, "chrNope", 1, 10) == []
    # a larger synthetic table: many blocks, several contigs
    big = ["chr%d\t%d\t%d\tDEL\tid%d" % (c, p, p + 50 + (p % 700), p) for c in (1, 2, 10) for p in range(5, 3000000, 1371)]
    workflow.write_bed_gz_with_index(str(tmp_path / "big.bed.gz"), workflow.sort_rows(big))
    assert gzip.open(str(tmp_path / "big.bed.gz"), "rt").read().splitlines() == workflow.sort_rows(big)
    for chrom, a, b in (("chr2", 100000, 120000), ("chr10", 1, 5000), ("chr1", 2999000, 3100000)):
        exp = [r for r in workflow.sort_rows(big) if r.split("\t")[0] == chrom and int(r.split("\t")[1]) < b and int(r.split("\t")[2]) > a - 1]
        assert workflow.tabix_query(str(tmp_path / "big.bed.gz"), chrom, a, b) == exp


def _world_to_files(world, d):
    return synth.write_world_files(world, str(d))


@pytest.mark.parametrize("case", [c for c in LOCUS if not [p for p in c["per_locus"] if "error" in p["scores"]]][:3],
                         ids=lambda c: c["name"])
def test_bed_cli_from_files_without_samtools(fake, case, tmp_path):
    """The same table from files on disk: reference windows through the .fai index, reads through the
    in-process BGZF/BAM/BAI reader - no samtools process anywhere.  A BAM is coordinate-sorted,
    so the expectation is the in-memory run over the same reads in coordinate order."""
    world = synth.world_from_json(case["world"])
    for c in world.reads:
        world.reads[c] = sorted(world.reads[c], key=lambda r: r.pos)          # stable: ties keep their order
    bed = tmp_path / "in.bed"
    bed.write_text(case["bed"])

    def run(ref, bam, out):
        assert cli.main(["bed", "--sv-input", str(bed), "--reference", ref, "--pacbio-input", bam,
                         "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]) == 0
        return out.read_text()

    seqio.set_backend(seqio.MemorySamtools(world))
    exp = run("ref.fa", "x.bam", tmp_path / "mem.vapor")
    fa, bam = _world_to_files(world, tmp_path)
    seqio.set_backend(seqio.InProcessBam())
    try:
        got = run(fa, bam, tmp_path / "files.vapor")
    finally:
        seqio.set_backend(None)
    assert got == exp and len(exp.splitlines()) == len(case["vapor_text"].splitlines())


def test_loci_started_on_threads_give_the_same_table(fake, tmp_path, monkeypatch):
    """pipeline.run_batch starts the loci of a batch on several threads when the backend reads files in-process (the
    BGZF inflation releases the GIL): the table is the one a single thread writes, row for row."""
    world = synth.make_world(seed=23, n_loci=40, svtypes=("DEL", "INV", "INS", "DEL"), span_range=(100, 1500), read_len=2500,
                             n_reads=6)
    for c in world.reads:
        world.reads[c] = sorted(world.reads[c], key=lambda r: r.pos)
    bed = tmp_path / "in.bed"
    bed.write_text(synth.bed_text(world))
    fa, bam = _world_to_files(world, tmp_path)

    def run(threads, out):
        monkeypatch.setenv("VAPOR_PREFETCH_THREADS", str(threads))
        seqio.set_backend(seqio.InProcessBam())
        try:
            assert cli.main(["bed", "--sv-input", str(bed), "--reference", fa, "--pacbio-input", bam,
                             "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]) == 0
        finally:
            seqio.set_backend(None)
        return out.read_text()

    one = run(1, tmp_path / "one.vapor")
    many = run(6, tmp_path / "many.vapor")
    assert many == one and len(one.splitlines()) >= 40


def test_native_cigar_walk_equals_the_python_statement():
    """vapor_cigar2alignstart (host helper in the C ABI) against the Python restatement of SF:309-337: the survey's
    known answers and random CIGARs with every operation, junk characters, empty strings and long walks."""
    import random
    assert seqio.cigar2alignstart_by_pos("100S500M20I300M", 1000, 1200, 2200) == [300, 0]
    assert seqio.cigar2alignstart_by_pos("50M300D500M", 1000, 1200, 2200) == [50, 150]
    with pytest.raises(IndexError):
        seqio.cigar2alignstart_by_pos("", 5, 9, 20)
    rnd = random.Random(11)
    for _ in range(3000):
        nops = rnd.choice([0, 1, 3, 30, 400, 2500])
        cig = "".join("%d%s" % (rnd.randint(1, rnd.choice([3, 40, 900])), rnd.choice("MMMMIDS=XNHPQ ")) for _ in range(nops))
        a = rnd.randint(1, 5000)
        st = a + rnd.randint(-50, rnd.choice([10, 500, 20000]))
        try:
            exp = seqio._cigar2alignstart_py(cig, a, st, st + 100)
        except IndexError:
            exp = "IndexError"
        try:
            got = seqio.cigar2alignstart_by_pos(cig, a, st, st + 100)
        except IndexError:
            got = "IndexError"
        assert got == exp, (cig[:60], a, st)


def test_queue_edge_case_design():
    """The crafted pair of tests/queue_case.py really asks the join's queue for 127, 129, 129 candidates (model of
    the device hash on the CPU; the GPU test compares the kernel's dots with the oracle)."""
    import queue_case
    read, allele, km = queue_case.build()
    assert queue_case.candidate_totals(read, allele)[:9] == [127, 0, 0, 0, 129, 0, 0, 0, 129]
    assert allele.count(km["P"]) == 2 and allele.count(km["Q"]) == 3 and allele.count(km["R"]) == 3


def test_deep_loci_host_finish_vs_reference(oracle):
    """The 33-64-read loci of tests/golden/deep_loci.json.gz: oracle statistics -> host finish (vapor_amd.finish)
    == the reference's per-read scores, result_organize_ins and gt_estimate_log_likelihood."""
    import deep_cases as dc
    from vapor_amd import finish
    seqs, rows, table, _n = dc.build()
    fe = FakeEngine(oracle)
    st = fe.score(fe.seqset(seqs), fe.make_pairs(rows))
    fin = {0: None, 1: finish.score_abs_dis_m1b, 3: finish.score_directed_dis_m1b_redefine_diagnal}
    for li, c in enumerate(dc.DEEP):
        scores, qs, gs, gt, gq = dc.expected(c)
        got = []
        for r in np.flatnonzero(table["locus"] == li):
            a, b, lr, la = st[2 * r], st[2 * r + 1], int(table["len_ref"][r]), int(table["len_alt"][r])
            if table["kind"][r] == 0:
                x, y = finish.score_abs_dis_m1b(a, b, lr, la), finish.score_within_10Perc_m1b(a, b, lr, la)
                if 0 not in x and 0 not in y:
                    got.append(min([1 - float(x[1]) / float(x[0]), 1 - float(y[1]) / float(y[0])]))
                elif 0 not in x:
                    got.append(1 - float(x[1]) / float(x[0]))
                elif 0 not in y:
                    got.append(1 - float(y[1]) / float(y[0]))
            else:
                x = fin[int(table["kind"][r])](a, b, lr, la)
                if 0 not in x:
                    got.append(1 - float(x[1]) / float(x[0]))
        assert got == scores, c["name"]
        s_qs, s_gs, idx, s_gq = finish.locus_summary(got)
        assert (float(s_qs), float(s_gs), idx, float(s_gq)) == (qs, gs, gt, gq), c["name"]


def test_figure_specs_match_what_the_reference_plots(fake):
    import figure_cases
    figure_cases.check_specs()


def test_figure_requests_of_the_drivers(fake):
    import figure_cases
    figure_cases.check_driver_requests()


def test_pool_size_follows_the_cpu_quota_and_the_ranks_on_the_host(monkeypatch):
    """pipeline._prefetch_threads: VAPOR_PREFETCH_THREADS wins; otherwise the usable cores (affinity cut to the container's
    CPU quota) divided by the ranks torchrun started on this host, at most eight; one thread for small batches and for
    backends that are not thread-safe."""
    from vapor_amd import pipeline

    class Be:
        threads_ok = True
    seqio.set_backend(Be())
    try:
        monkeypatch.delenv("VAPOR_PREFETCH_THREADS", raising=False)
        monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
        monkeypatch.setattr(pipeline, "_usable_cores", lambda: 16)
        assert pipeline._prefetch_threads(1000) == 8
        assert pipeline._prefetch_threads(10) == 1
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "5")
        assert pipeline._prefetch_threads(1000) == 3
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "64")
        assert pipeline._prefetch_threads(1000) == 1
        monkeypatch.setenv("VAPOR_PREFETCH_THREADS", "6")
        assert pipeline._prefetch_threads(1000) == 6
        Be.threads_ok = False
        assert pipeline._prefetch_threads(1000) == 1
    finally:
        seqio.set_backend(None)
    monkeypatch.undo()
    assert 1 <= pipeline._usable_cores() <= (os.cpu_count() or 1)


@pytest.mark.parametrize("procs", ["0", "2"])
def test_figures_of_a_batch_are_drawn_by_worker_processes(fake, tmp_path, monkeypatch, procs):
    """The drivers' Figure requests of a batch go through figures.make_figures: one device pass for all their dot plots,
    the drawing in spawned worker processes (VAPOR_FIGURE_PROCS=2) or here (0); when run_batch returns every PNG the
    reference would have written is on disk, and the batched specifications are the single ones."""
    import figure_cases
    from vapor_amd import drivers, figures
    monkeypatch.setenv("VAPOR_FIGURE_PROCS", procs)
    figures.shutdown()
    cases = figure_cases.FIG["cases"]
    reqs = [drivers.Figure(c["scores"], c["best_read"], c["k"], c["ref_seq"], c["alt_seq"],
                           str(tmp_path / ("f%d.png" % n))) for n, c in enumerate(cases)]
    many = figures.figure_specs(reqs)
    for r, m in zip(reqs, many):
        one = figures.figure_spec(r)
        assert (one is None) == (m is None)
        if one is not None:
            assert one["name"] == m["name"] and all(np.array_equal(a["hits"], b["hits"]) and a["xticks"] == b["xticks"]
                                                    for a, b in zip(one["subplots"], m["subplots"]))
    try:
        figures.make_event_figure_1.batch(reqs)
        figures.make_event_figure_1.wait()
        drawn = sorted(f for f in os.listdir(tmp_path) if f.endswith(".png"))
        assert len(drawn) == sum(c["drawn"] is not None for c in cases) >= 6
        assert all(os.path.getsize(tmp_path / f) > 2000 for f in drawn)
        from vapor_amd import hostpool
        assert (hostpool._pool is not None) == (procs == "2")
        # what a drawing raises (here: a directory that does not exist) comes back to the caller, from a worker too
        bad = drivers.Figure(cases[0]["scores"], cases[0]["best_read"], cases[0]["k"], cases[0]["ref_seq"], cases[0]["alt_seq"],
                             str(tmp_path / "no_such_dir" / "x.png"))
        with pytest.raises((OSError, RuntimeError)):
            figures.make_event_figure_1.batch([bad])
            figures.make_event_figure_1.wait()
    finally:
        figures.shutdown()


def test_repeat_clustering_in_host_workers_equals_the_callers_own(fake, monkeypatch):
    """refine_windows hands the X-means clustering of the windows in the (0.1, 0.5) band to the host worker processes
    when there are eight or more: with a seed (the reference runs it unseeded) the answers are those of the calling
    process, slot for slot."""
    from vapor_amd import hostpool
    rng = np.random.default_rng(12)
    seqs = []
    for _ in range(10):
        a, b, c = (synth.random_dna(rng, int(n)) for n in (rng.integers(150, 260), rng.integers(260, 340), rng.integers(150, 260)))
        seqs.append(a + b + b + c)                      # a tandem duplication: its copy's dots lie off the diagonal
    seqs.append(synth.random_dna(rng, 500))             # and one window that is not in the band
    monkeypatch.setenv("VAPOR_QC_SEED", "7")
    hostpool.shutdown()
    monkeypatch.setenv("VAPOR_HOST_PROCS", "0")
    own = pipeline.refine_windows(fake, seqs)
    assert sum(1 for r in own if isinstance(r, list) and r[1] != "Error" and r[1][1] != [0]) >= 8
    monkeypatch.setenv("VAPOR_HOST_PROCS", "2")
    try:
        pooled = pipeline.refine_windows(fake, seqs)
        assert hostpool._pool is not None and len(hostpool._pool.procs) >= 1
    finally:
        hostpool.shutdown()
    assert repr(pooled) == repr(own)


def test_repeat_clustering_direct_routes_equal_the_public_calls():
    """repeat_qc's direct routes - the k-means++ / Lloyd routines KMeans.fit calls, and scipy.cluster.vq.kmeans's loop on its
    compiled routines - against the public calls they stand for (sklearn.cluster.KMeans(...).fit, scipy.cluster.vq.kmeans)
    with a seed: labels, centres and code books equal bit for bit over random point sets of the shapes a self dot plot gives
    (lines off the diagonal, noise, duplicates, a handful of points), and the X-means block sizes of whole windows equal
    either way.  In a child process with one OpenMP thread, as the host workers run."""
    import subprocess
    import sys
    code = r"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, %r)
from scipy.cluster.vq import whiten
from vapor_amd import repeat_qc as R
warnings.simplefilter("ignore")
assert R._direct_ok(), "the direct routes were refused by their self-check"
rng = np.random.default_rng(77)
n_fit = n_vq = 0
for t in range(160):
    n = int(rng.choice([1, 2, 3, 4, 5, 9, 30, 120, 400, 900]))
    kind = t %% 4
    if kind == 0:
        pts = np.column_stack((rng.integers(0, 5000, n), rng.integers(0, 5000, n)))
    elif kind == 1:                                  # a line off the diagonal plus noise
        i = rng.integers(1000, 1000 + 3 * n, n); pts = np.column_stack((i + 700, i))
        pts[: n // 5] = np.column_stack((rng.integers(0, 5000, n // 5), rng.integers(0, 5000, n // 5)))
    elif kind == 2:                                  # two blocks
        pts = np.column_stack((rng.integers(0, 300, n), rng.integers(0, 300, n))); pts[n // 2:] += 4000
    else:                                            # many duplicates
        pts = np.column_stack((rng.integers(0, 3, n) * 100, rng.integers(0, 3, n) * 100))
    ks = list(range(1, min(5, n + 1)))
    seed = int(rng.integers(0, 1000))
    for a, b in zip(R._kmeans_fits_public(pts, ks, seed), R._kmeans_fits_direct(pts, ks, seed)):
        assert np.array_equal(a.labels_, b.labels_) and np.array_equal(a.cluster_centers_, b.cluster_centers_) and a.n_clusters == b.n_clusters, (t, n)
        n_fit += 1
    if pts[:, 0].std() > 0 and pts[:, 1].std() > 0:
        w = whiten(pts)
        for k in ks[1:]:
            assert np.array_equal(R._vq_kmeans_public(w, k, seed), R._vq_kmeans_direct(w, k, seed)), (t, n, k)
            n_vq += 1
os.environ["VAPOR_QC_SEED"] = "7"
wins = []
for span in (120, 400, 900, 2500):
    i = np.arange(3000, 3000 + span)
    nj, ni = rng.integers(0, 12000, 150), rng.integers(0, 12000, 150)
    keep = nj > ni
    wins.append(np.column_stack((np.concatenate((i + span, nj[keep])), np.concatenate((i, ni[keep])))))
direct = [R.cluster_sizes_of_points(w) for w in wins]
R._DIRECT = False
public = [R.cluster_sizes_of_points(w) for w in wins]
assert repr(direct) == repr(public) and any(len(d) > 1 for d in direct)
print("fits", n_fit, "code books", n_vq, "windows", len(wins))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fits" in r.stdout, r.stderr[-3000:]


def test_kept_figure_writes_the_same_bytes_as_a_fresh_one(fake, tmp_path):
    """figures.render keeps its matplotlib figure between calls; figures.render_fresh makes the reference's calls one by
    one on a new figure: the PNG files are equal byte for byte, whatever was drawn before (larger, smaller, other ticks)."""
    import figure_cases
    from vapor_amd import drivers, figures
    cases = figure_cases.FIG["cases"]
    reqs = [drivers.Figure(c["scores"], c["best_read"], c["k"], c["ref_seq"], c["alt_seq"], "x.png") for c in cases]
    specs = [s for s in figures.figure_specs(reqs) if s is not None]
    rng = np.random.default_rng(9)
    for n in (40, 30000, 700):                             # and synthetic ones of very different extents
        d = np.arange(n, dtype=np.int32)
        h = np.stack([d, d], 1)
        h = np.concatenate([h, rng.integers(0, n, (n // 20 + 3, 2)).astype(np.int32)])
        h = h[np.lexsort((h[:, 1], h[:, 0]))]
        t = figures.x_ticks(int(h[:, 0].max()))
        specs.append({"name": "x.png", "subplots": [{"pos": p, "title": ti, "hits": h, "xticks": t, "xticklabels": [str(i) for i in t]}
                                                    for p, ti in zip(figures.POSITIONS, figures.TITLES)]})
    assert len(specs) >= 9
    for n, spec in enumerate(specs + specs[::-1]):
        a, b = dict(spec, name=str(tmp_path / ("kept%d.png" % n))), dict(spec, name=str(tmp_path / ("fresh%d.png" % n)))
        figures.render(a)
        figures.render_fresh(b)
        assert open(a["name"], "rb").read() == open(b["name"], "rb").read(), n



def test_worker_that_dies_inside_its_reply_is_a_lost_worker():
    """a worker that ends after the tag byte left a short header (struct.error) or a truncated pickle, which
    callers stored as the locus' result; it is WorkerLost now (the callers' fallback), the process is dropped, the next task
    gets a new one, and close() does not wait for ever."""
    from vapor_amd import hostpool
    w = hostpool.Workers(1)
    try:
        assert w.submit("operator", "add", 2, 3).result(timeout=60) == 5
        # the worker writes a tag byte and three bytes of the length to its reply stream, then ends
        with pytest.raises(hostpool.WorkerLost):
            w.submit("builtins", "exec", "import os\nos.write(1, b'\\x00\\x01\\x02\\x03')\nos._exit(3)").result(timeout=60)
        # ... a whole header that promises more than comes
        with pytest.raises(hostpool.WorkerLost):
            w.submit("builtins", "exec", "import os, struct\nos.write(1, b'\\x00' + struct.pack('<q', 500) + b'abc')\nos._exit(3)").result(timeout=60)
        # ... and a body that is no pickle
        with pytest.raises(hostpool.WorkerLost):
            w.submit("builtins", "exec", "import os, struct\nos.write(1, b'\\x00' + struct.pack('<q', 3) + b'abc')\nos._exit(3)").result(timeout=60)
        assert w.submit("operator", "mul", 4, 5).result(timeout=60) == 20
        assert len(w.procs) == 4
    finally:
        procs = list(w.procs)
        w.close()
    assert all(p.poll() is not None for p in procs)


def test_memory_backend_native_chop_equals_the_per_record_statement(monkeypatch):
    """MemorySamtools.chop (one vapor_chop_records call per region over the contig's records as arrays) against the per-record
    statement of chop_pacbio_read_by_pos (SF:339-354) it replaces in the synthetic worlds: same reads, offsets and miss_bp for
    random regions, clipped / inserted / deleted CIGAR starts (the SURVEY's known answers among them), and the IndexError of
    a record without CIGAR."""
    from vapor_amd import seqio
    w = synth.make_world(seed=91, n_loci=6, svtypes=("DEL", "TANDUP", "INV", "INS"), span_range=(150, 2500), read_len=5000, n_reads=14)
    chrom = "c1"
    base = w.contigs[chrom]
    extra = [("k1", 1000, "100S500M20I300M"), ("k2", 1000, "50M300D500M"), ("k3", 900, "10I400M5D400M"), ("k4", 1190, "3S20M2000D900M"),
             ("k5", 1200, "900M"), ("k6", 1201, "900M"), ("k7", 700, "1000=")]
    for name, pos, cg in extra:
        n_q = sum(int(n) for n, c in __import__("re").findall(r"(\d+)([MIDNSHP=X])", cg) if c in "SIM=")
        n_r = sum(int(n) for n, c in __import__("re").findall(r"(\d+)([MIDNSHP=X])", cg) if c in "MD=")
        w.reads[chrom].append(synth.SamRecord(name, chrom, pos, cg, synth.random_dna(np.random.default_rng(pos), n_q), n_r))
    be = seqio.MemorySamtools(w)
    rng = np.random.default_rng(5)
    n_kept = 0
    for _ in range(300):
        c = "c%d" % int(rng.integers(1, 7))
        a = int(rng.integers(1, 3000)); wd = int(rng.integers(1, 2500)); f = int(rng.choice([0, 1, 50, 200, 500]))
        monkeypatch.setenv("VAPOR_MEMORY_CHOP", "records")
        ref = be.chop("x.bam", c, a, a + wd, f)
        monkeypatch.delenv("VAPOR_MEMORY_CHOP")
        assert be.chop("x.bam", c, a, a + wd, f) == ref, (c, a, wd, f)
        n_kept += len(ref)
    assert n_kept > 300
    # the known answers of cigar2alignstart_by_pos through the batch helper
    got = {r[2]: r for r in be.chop("x.bam", chrom, 1200, 1400, 500)}
    assert got["k1"][1] == 0 and got["k2"][1] == 150 and "k6" not in got and "k5" in got
    w.reads[chrom].append(synth.SamRecord("nocigar", chrom, 1100, "*", "ACGT" * 300, 1200))
    be2 = seqio.MemorySamtools(w)
    with pytest.raises(IndexError):
        be2.chop("x.bam", chrom, 1200, 1400, 500)


def test_score_jobs_relaxes_the_collector_and_puts_it_back(monkeypatch):
    """cli.score_jobs raises the cyclic collector's thresholds for the duration of the scoring (its passes over everything a
    run keeps alive cost a quarter of an 8 000-locus run) and restores what it found - also when the scoring raises."""
    import gc
    seen = []
    was = gc.get_threshold()

    def inner(jobs, chunk, figure_fn, t0):
        seen.append(gc.get_threshold())
        if jobs == "boom":
            raise ValueError("x")
        return []
    monkeypatch.setattr(cli, "_score_jobs", inner)
    assert cli.score_jobs([], 4) == [] and gc.get_threshold() == was
    with pytest.raises(ValueError):
        cli.score_jobs("boom", 4)
    assert gc.get_threshold() == was
    assert all(t[0] >= 200000 and t[1] >= 50 and t[2] >= 1000 for t in seen) and len(seen) == 2
    gc.set_threshold(300000, 60, 2000)                       # (a caller's own, higher thresholds are kept)
    try:
        cli.score_jobs([], 4)
        assert seen[-1] == (300000, 60, 2000) and gc.get_threshold() == (300000, 60, 2000)
    finally:
        gc.set_threshold(*was)


@pytest.mark.parametrize("in_flight", ["1", "2"])
def test_bed_cli_chunks_in_flight(fake, in_flight, tmp_path, monkeypatch):
    """cli.score_jobs with several chunks (--chunk 3 on the eight-locus world), one or two of them in flight on threads:
    the reference's table byte for byte either way (the tests' stand-in engine is shared by the threads)."""
    case = [c for c in LOCUS if c["name"] == "bed_small_mix"][0]
    monkeypatch.setenv("VAPOR_CHUNKS_IN_FLIGHT", in_flight)
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
    bed = tmp_path / "in.bed"
    bed.write_text(case["bed"])
    out = tmp_path / "out.vapor"
    assert cli.main(["bed", "--sv-input", str(bed), "--reference", "ref.fa", "--pacbio-input", "x.bam", "--chunk", "3",
                     "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]) == 0
    assert out.read_text() == case["vapor_text"]
    assert len(fake.batches) >= 3            # three chunks, a few plans each


def test_output_rows_of_a_table_equal_the_reference_named_composition():
    """cli.output_rows / finish.row_tails - the whole table's tails from one call of the library's host helper
    (vapor_row_tails: rounding, Rec strings, mean of the positive scores, counts) - against result_organize_ins (SF:1219-1231)
    -> format_output_row with gt_estimate_log_likelihood (SF:2054-2088) per locus: ties of round(s, 2) (0.125, 2.675, 1.005,
    x.xx5 in general), scores that round to 0.0 / -0.0, integers, long lists (numpy's pairwise blocks), loci the helper hands
    back (NaN, inf, huge), empty lists, a head field that reads NA; and the helper's argument checks."""
    import ctypes
    import random
    from vapor_amd import _lib, cli, finish
    from vapor_amd import simple_function as SF
    rng = random.Random(11)
    special = [0.0, -0.0, 0.005, -0.005, 0.125, 0.375, 2.675, 1.005, -1.005, 0.015, 0.025, 0.035, 0.045, 1e-9, -1e-9, 0.004999999,
               0.00500000001, 1.0, -1.0, 10.0, 100.5, 0.995, 0.9949999, 3, -2, np.float64(0.25)]
    heads, tables = [], []
    for t in range(6000):
        n = rng.choice([0, 1, 2, 3, 7, 8, 9, 15, 16, 17, 20, 20, 20, 40, 127, 128, 129, 300])
        sc = []
        for _ in range(n):
            r = rng.random()
            if r < .1:
                sc.append(rng.choice(special))
            elif r < .2:
                sc.append((rng.randint(-10 ** 5, 10 ** 5) + .5) / 100.0)
            elif r < .25:
                sc.append(rng.uniform(-1e12, 1e12))
            else:
                sc.append(1 - rng.expovariate(1.0) * rng.choice([.1, 1, 10]))
        if n and rng.random() < .03:
            sc[rng.randrange(n)] = rng.choice([float("nan"), float("inf"), -float("inf"), 1e13, -3e15, 1e300])
        heads.append(["c%d" % t, "100", "200", rng.choice(["DEL", "NA", "sv1"])])
        tables.append(sc)
    with np.errstate(divide="ignore"):
        lines, tails = cli.output_rows(heads, tables)
        for head, sc, line, tail in zip(heads, tables, lines, tails):
            key = ":".join(head[:3])
            assert line == SF.format_output_row(key.split(":") + [head[3]] + finish.result_organize_ins([key, sc])[1:]), sc
            exp = finish.row_tail(sc)
            assert [str(x) for x in tail] == [str(x) for x in exp] and [type(x) for x in tail] == [type(x) for x in exp]
        vcf_lines = cli.output_rows([[":".join(h[:3])] for h in heads[:500]], tables[:500])[0]
        for head, sc, line in zip(heads, tables, vcf_lines):
            assert line == SF.format_output_row(finish.result_organize_ins([":".join(head[:3]), sc]))
    assert cli.output_rows([], []) == ([], [])
    odd = [[0.5], ["0.25", "-1"], (0.125, 7)]                                     # (numerals as text, a tuple)
    assert finish.row_tails(odd) == [finish.row_tail(sc) for sc in odd]
    with pytest.raises((TypeError, ValueError)):
        finish.row_tails([[0.5], None])                                          # as row_tail(None) raises
    lib = _lib.load()
    off = np.array([0, 2, 1], dtype=np.int64)
    one = np.zeros(4)
    i32 = np.zeros(4, dtype=np.int32)
    to = np.zeros(4, dtype=np.int64)
    buf = ctypes.create_string_buffer(64)
    assert lib.vapor_row_tails(2, off.ctypes.data, one.ctypes.data, one.ctypes.data, i32.ctypes.data, i32.ctypes.data,
                               ctypes.addressof(buf), 64, to.ctypes.data) == _lib.E_ARG
    assert lib.vapor_row_tails(1, None, None, None, None, None, None, 0, None) == _lib.E_ARG
    off = np.array([0, 3], dtype=np.int64)
    sc = np.array([0.5, -1.25, 123456.789])
    assert lib.vapor_row_tails(1, off.ctypes.data, sc.ctypes.data, one.ctypes.data, i32.ctypes.data, i32.ctypes.data,
                               ctypes.addressof(buf), 4, to.ctypes.data) == _lib.E_OVERFLOW
    assert to[1] == len("0.5,-1.25,123456.79")


def test_output_row_equals_the_reference_named_composition():
    """cli.output_row / finish.row_tail (one rounding per score, numpy's summation order restated) against the composition it
    replaces - result_organize_ins (SF:1219-1231) -> format_output_row with gt_estimate_log_likelihood (SF:2054-2088) - on
    random score lists: empty, short and long ones, scores that round to 0.0 / -0.0, a head field that reads NA."""
    import random
    from vapor_amd import cli, finish
    from vapor_amd import simple_function as SF
    rng = random.Random(7)
    for t in range(4000):
        n = rng.choice([0, 1, 2, 5, 7, 8, 9, 15, 16, 17, 20, 20, 20, 33, 64, 127, 128, 140])
        sc = []
        for _ in range(n):
            r = rng.random()
            sc.append(rng.choice([r, -r * 30, 0.004999, 0.005, 0.0050001, -0.004, 0.0, 1.0 - r * 1e-9, np.float64(r)]))
        head = ["c%d" % t, "100", "200", rng.choice(["DEL", "NA", "sv1"])]
        key = ":".join(head[:3])
        old = SF.format_output_row(key.split(":") + [head[3]] + finish.result_organize_ins([key, sc])[1:])
        assert cli.output_row(key.split(":") + [head[3]], sc)[0] == old, (t, sc)
        old_vcf = SF.format_output_row(finish.result_organize_ins([key, sc]))
        assert cli.output_row([key], sc)[0] == old_vcf
    for _ in range(20000):
        a = [rng.random() ** rng.randint(1, 5) for _ in range(rng.randint(1, 150))]
        assert finish._mean_like_numpy(a) == np.mean(a) and str(finish._mean_like_numpy(a)) == str(np.mean(a))


def test_alleles_travel_as_descriptors_not_as_bytes(fake, tmp_path):
    """the drivers' alt alleles are slices of the window they have read (drivers.Allele), and the executor
    hands the segments to the library (SeqSet `derived`) instead of the bytes; the str.upper() twins of abs_dis_m1b
    (SF:183-184) of a soft-masked window are descriptors too - no window is uploaded twice."""
    from vapor_amd import drivers
    case = [c for c in LOCUS if c["name"] == "bed_small_mix"][0]
    seqio.set_backend(seqio.MemorySamtools(synth.world_from_json(case["world"])))
    bed = tmp_path / "in.bed"
    bed.write_text(case["bed"])
    out = tmp_path / "out.vapor"
    seen = []
    real = fake.seqset

    def spy(seqs, upper=None, derived=None):
        seen.append((list(seqs), derived))
        return real(seqs, upper, derived)
    fake.seqset = spy
    assert cli.main(["bed", "--sv-input", str(bed), "--reference", "ref.fa", "--pacbio-input", "x.bam",
                     "--output-path", str(tmp_path / "figs"), "--output-file", str(out), "--no-figures"]) == 0
    assert out.read_text() == case["vapor_text"]
    scored = [(s, d) for s, d in seen if d]
    assert scored, "no sequence set carried derived sequences"
    n_derived = sum(len(d) for _s, d in scored)
    assert n_derived >= 8                                   # every locus of the eight has an alt allele built from its window
    for seqs, der in scored:
        assert len(set(map(id, seqs))) == len(seqs)         # no string uploaded twice
        for segs, _up in der:
            assert all(0 <= p < len(seqs) and o >= 0 and o + n <= len(seqs[p]) for p, o, n, _rc in segs)
    # the segment lists say what the strings say
    a = drivers._cat(("ACGTTGCAAC", None, 4), ("ACGTTGCAAC", 2, -3, True), ("ACGTTGCAAC", -4, None))
    assert a == "ACGT" + seqio.reverse(seqio.complementary("GTTGC")) + "CAAC" and [(o, n, rc) for _p, o, n, rc in a.segs] == [(0, 4, False), (2, 5, True), (6, 4, False)]
    b = drivers._cat(("ACGTRGCAAC", 2, 8, True))             # complementary() drops the R: not a slice any more
    assert b == seqio.reverse(seqio.complementary("GTRGCA")) and len(b) == 5 and b.segs is None
    # a soft-masked deletion: four allele sequences (ref, alt and their upper twins), one window uploaded
    ref = "acgt" * 30 + synth.random_dna(np.random.default_rng(2), 1400)
    alt = drivers._cat((ref, None, 500), (ref, -500, None))
    reads = [[ref[:900], 0, "r%d" % t] for t in range(4)]
    del seen[:]
    pipeline.score_requests(fake, [drivers.Score("del", ref, alt, reads, 10)])
    (seqs, der), = seen
    assert seqs.count(ref) == 1 and alt not in seqs and len(seqs) == 1 + len(reads)
    assert sorted((len(sg), up) for sg, up in der) == [(1, True), (2, False), (2, True)]


def test_three_threads_into_a_cold_repeat_check():
    """the hang of gpurun_out/r4_hang.log (three chunk threads in concurrent first imports of scikit-learn /
    SciPy inside repeat_qc; fixed in 8596378 by repeat_qc._warm) has a regression test: a fresh interpreter, three threads
    into a cold cluster_sizes at once, finished within the timeout, never more than one thread inside a first import of the
    clustering libraries - and without the fix (the negative control) the same harness sees all three in there together."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, VAPOR_QC_SEED="7", HANG_CASE_SECONDS="100")
    case = os.path.join(ROOT, "tests", "hang_case.py")
    r = subprocess.run([sys.executable, case], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["threads_importing_at_once"] == 1, got
    assert got["sizes"][0] == got["sizes"][1] == got["sizes"][2] and len(got["sizes"][0]) >= 1
    r = subprocess.run([sys.executable, case, "nowarm"], capture_output=True, text=True, timeout=120, env=env)
    if r.returncode == 0:                                    # (without the fix it may also hang: the dump timer then ends it)
        assert json.loads(r.stdout.strip().splitlines()[-1])["threads_importing_at_once"] > 1


def test_sort_rows_is_gnu_sort_version_order():
    """ConcatVaPoR's `sort -Vk1,1 -k2,2n -k3,3n` (wdl/TasksBenchmark.wdl:286-301) is pinned on the real
    thing - coreutils' sort (LC_ALL=C) over a table of 12 000 contigs with the names genomes really carry (chr2_alt,
    chrUn_*, *_random, GL000207.1, HLA-A*01:01, leading zeros, suffixes, '~') - not on workflow's own reader."""
    import random
    import shutil
    import subprocess
    from vapor_amd import workflow
    if not shutil.which("sort"):
        pytest.skip("no coreutils sort")
    random.seed(5)
    names = ["chr%d" % i for i in range(1, 23)] + [
        "chrX", "chrY", "chrM", "chrEBV", "chr2_alt", "chr2_KI270715v1_random", "chr10_alt", "chr1_alt", "chrUn_KI270302v1", "chrUn_GL000195v1",
        "chrUn_KI270302v2", "chr11_KI270721v1_random", "GL000207.1", "KI270728.1", "HLA-A*01:01:01:01", "HLA-DRB1*15:01:01:01", "1", "2", "10",
        "X", "MT", "chr01", "chr1a", "chr1_", "chr1-2", "chr1~x", "chr1.fa", "chr1.2.fa", "Chr1", "chr", "chr00", "chr1.10", "chr1.9", "chr1.fa.gz",
        "chr1.fa~", ".hidden", ".hidden2", "chr1.a1", "chr1.1a", "a.b.c", "a.b", "a", "a.0", "a.00", "a.b~", "000", "0", "00x", "x00", "x0", "x"]
    names += ["ctg%06d" % random.randrange(10 ** 6) for _ in range(6000)] + ["scaf_%d.%d" % (random.randrange(300), random.randrange(30)) for _ in range(3000)]
    names += ["chrUn_%s%06dv%d" % (random.choice(["KI", "GL", "JH"]), random.randrange(10 ** 6), random.randrange(1, 4)) for _ in range(3000)]
    assert len(set(names)) > 11000
    rows = []
    for n in names:
        for _ in range(random.choice([1, 1, 2, 3])):
            a = random.randrange(0, 10 ** 8)
            rows.append("%s\t%d\t%d\tDEL\t0.5" % (n, a, a + random.randrange(1, 10 ** 5)))
    rows += ["chr1\t100\t200\tDEL\t1", "chr1\t100\t200\tDEL\t0", "chr1\t100\t150\tINV\t1", "chr1\t99\t1000\tINV\t1", "chr1\t0100\t120\tX\t1",
             "chr1\t-5\t10\tX\t1", "chr1\t1e3\t10\tX\t1", "chr1\t12.5\t10\tX\t1", "chr1\t.5\t10\tX\t1", "chr1\tNA\t10\tX\t1"]
    random.shuffle(rows)
    want = subprocess.run(["sort", "-Vk1,1", "-k2,2n", "-k3,3n"], input="\n".join(rows) + "\n", capture_output=True, text=True,
                          env=dict(os.environ, LC_ALL="C"), check=True).stdout.splitlines()
    got = workflow.sort_rows(rows)
    bad = [t for t in range(len(want)) if got[t] != want[t]]
    assert not bad, (len(bad), [(got[t][:40], want[t][:40]) for t in bad[:5]])
    # and name against name, both ways round
    some = names[:80]
    srt = subprocess.run(["sort", "-V"], input="\n".join(sorted(set(some))) + "\n", capture_output=True, text=True, env=dict(os.environ, LC_ALL="C"), check=True).stdout.splitlines()
    assert sorted(set(some), key=workflow.version_key) == srt


def test_worlds_drawn_from_the_truth_sets_span_distribution():
    """synth.make_world(span_dist="simulate") samples the type-wise span distribution of the reference's
    simulated truth sets (vapor_amd/data/simulate_spans.json, written from simulate/Structural_Variants_het by
    oracle/gen_span_dist.py): 50 bp - 100 kb, median ~2.8 kb, 7-10 % of the deletions and inversions >= 10 kb, tandem duplications
    below 5 kb, insertion lengths of the mobile elements."""
    tb = synth.simulate_span_tables()
    assert tb["simple"]["DEL"]["n"] == 1846 and tb["simple"]["INV"]["n"] == 447 and tb["simple"]["TANDUP"]["n"] == 617
    assert tb["simple"]["DEL"]["min"] >= 50 and 99000 < tb["simple"]["DEL"]["max"] <= 100000 and tb["simple"]["TANDUP"]["max"] < 5000
    assert 2500 < tb["simple"]["DEL"]["median"] < 3100 and 0.08 < tb["simple"]["DEL"]["frac_ge_10kb"] < 0.11
    assert len(tb["simple"]["DEL"]["quantiles"]) == 201 and tb["insertion_length"]["n"] > 200
    w = synth.make_world(seed=5, n_loci=1500, svtypes=("DEL", "DEL", "TANDUP", "INV", "INS"), read_len=600, n_reads=1, span_dist="simulate")
    sp = {t: np.array([l.end - l.start for l in w.loci if l.svtype == t]) for t in ("DEL", "INV", "TANDUP")}
    for t in ("DEL", "INV"):
        assert 2200 < np.median(sp[t]) < 3500 and 0.05 < (sp[t] >= 10000).mean() < 0.14 and sp[t].min() >= 50 and sp[t].max() > 50000, t
    assert sp["TANDUP"].max() < 5000 and 1900 < np.median(sp["TANDUP"]) < 3100
    ins = np.array([len(l.ins_seq) for l in w.loci if l.svtype == "INS"])
    assert ins.min() >= 40 and ins.max() <= 6100 and 200 < np.median(ins) < 1200
    # seeded: the same world again
    w2 = synth.make_world(seed=5, n_loci=40, svtypes=("DEL", "DEL", "TANDUP", "INV", "INS"), read_len=600, n_reads=1, span_dist="simulate")
    assert [(l.start, l.end) for l in w2.loci] == [(l.start, l.end) for l in w.loci[:40]]
