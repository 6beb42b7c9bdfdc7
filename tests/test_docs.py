"""The documents say what the sources do: every environment switch the library reads is listed in DESIGN.md's table
(VERDICT r02 "weak" 11: run-time switches must at least be accounted for), and every entry point of the headers is
named in INTEGRATION.md or DESIGN.md."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "locations-recommender_amd", "csrc")


def read(*parts):
    with open(os.path.join(ROOT, *parts)) as f:
        return f.read()


def test_every_environment_switch_is_documented():
    design = read("DESIGN.md")
    table = design[design.index("### Environment switches"):design.index("## 8. Status and numbers")]
    seen = set()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h", ".cpp")):
            seen |= set(re.findall(r'(?:getenv|debug_env)\("(LOCREC_[A-Z0-9_]+)"', read("locations-recommender_amd", "csrc", name)))
    for name in sorted(os.listdir(os.path.join(ROOT, "locations-recommender_amd"))):
        if name.endswith(".py"):
            seen |= set(re.findall(r'environ(?:\.get)?[\[(]"(LOCREC_[A-Z0-9_]+)"', read("locations-recommender_amd", name)))
    assert len(seen) > 40
    missing = sorted(s for s in seen if s not in table)
    assert not missing, f"switches read by the sources but absent from DESIGN.md's table: {missing}"


def test_every_entry_point_is_named_in_the_documents():
    docs = read("DESIGN.md") + read("INTEGRATION.md") + read("README.md")
    names = set()
    for header in ("locrec.h", "locrec_parquet.h"):
        names |= set(re.findall(r"\b(locrec_[a-z0-9_]+)\s*\(", read("include", header)))
    names = {n for n in names if not n.endswith("_t")}
    assert len(names) > 80
    # DESIGN.md section 1's table abbreviates families ("locrec_knn_create/destroy", "*_set_stream",
    # "locrec_parquet_read_knn/read_edges/free_*"): a name counts as documented when it is spelled out somewhere, or when
    # its part behind the family prefix stands in that table as a word (or as a `word_*` family)
    design = read("DESIGN.md")
    table = design[design.index("**Entry points**"):design.index("**Host language.**")]
    words = set(re.findall(r"[a-z0-9_]+", table))
    stars = {w for w in re.findall(r"([a-z0-9_]+_)\*", table)}

    def documented(n):
        if n in docs:
            return True
        for fam in ("locrec_knn_replicas_", "locrec_sg_sharded_", "locrec_sg_group_", "locrec_sg_shard_", "locrec_knn_",
                    "locrec_sg_", "locrec_cache_", "locrec_parquet_", "locrec_"):
            if n.startswith(fam):
                tail = n[len(fam):]
                if tail in words or any(tail.startswith(st) for st in stars):
                    return True
        return False

    missing = sorted(n for n in names if not documented(n))
    assert not missing, f"entry points no document mentions: {missing}"
