"""Parses the key set of aux_info/meta_info.json and the file names of aux_info out of the reference's GZipWriter::writeMeta
(/root/reference/src/output/GZipWriter.cpp) into tests/golden/meta_info_keys.json — the list tests/test_outputs.py diffs the product's files against.
Run in the build container (the reference tree does not exist on the GPU box)."""
import json, os, re, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse(ref=REF):
    src = open(os.path.join(ref, "src", "output", "GZipWriter.cpp")).read()
    a = src.index("bool GZipWriter::writeMeta(")
    i = src.index("{", a); depth = 0; b = i
    for b in range(i, len(src)):                        # the function's body by brace matching (no braces in its string literals)
        if src[b] == "{": depth += 1
        elif src[b] == "}":
            depth -= 1
            if depth == 0: break
    body = src[i:b + 1]
    body_nc = re.sub(r"/\*.*?\*/", "", body, flags=re.S)                       # the commented-out expected_gc / observed_gc block
    keys = list(dict.fromkeys(re.findall(r'make_nvp\(\s*"([A-Za-z0-9_]+)"', body_nc)))    # keep_duplicates sits in two switch arms
    files = re.findall(r'auxDir\s*/\s*"([A-Za-z0-9_.]+)"', body_nc)
    always = [f for f in files if f in ("fld.gz", "expected_bias.gz", "observed_bias.gz", "observed_bias_3p.gz", "meta_info.json")]
    return {"source": "src/output/GZipWriter.cpp GZipWriter::writeMeta", "keys": keys, "aux_files": files, "aux_files_always": always,
            "aux_files_seq_bias": [f for f in files if f.endswith("_seq.gz")], "aux_files_gc_bias": [f for f in files if f.endswith("_gc.gz")],
            "aux_files_pos_bias": [f for f in files if f.endswith("_pos.gz")]}


if __name__ == "__main__":
    out = parse()
    json.dump(out, open(os.path.join(HERE, "meta_info_keys.json"), "w"), indent=1)
    print(len(out["keys"]), "keys,", len(out["aux_files"]), "files")
