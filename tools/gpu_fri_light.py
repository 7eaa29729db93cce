"""First-light for Merkle + FRI on the GPU: golden roots / transcripts through the Python mirror."""
import os, sys, json, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stark_brainfuck_amd as sb
from oracle import ref_oracle as o

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SEED = 0x5EED
XF = sb.ExtensionField.main()
BFI = XF.modulus.coefficients[0].field

m = json.load(open(os.path.join(G, "merkle.json")))
for t in m["xfe_trees"]:
    n = t["n"]
    leaves = [XF.from_limbs([o.felt(SEED + t["seed_offset"], 3 * i + k) for k in range(3)]) for i in range(n)]
    tree = sb.Merkle(leaves)
    print("merkle", n, tree.root().hex() == t["root"], [x.hex() for x in tree.nodes] == t["nodes"],
          all([x.hex() for x in tree.open(i)] == t["paths"][i] for i in range(n)), flush=True)

f = json.load(open(os.path.join(G, "fri.json")))
for tag in ["d16_t2", "d64_t8", "d1024_t4", "test_fri_valid", "test_fri_disturbed", "d16_t2_prepushed"]:
    rec = f[tag]
    d = 1 << rec["log_degree"]
    if tag.startswith("test_fri"):
        coeffs = np.zeros((3, d), dtype=np.uint64); coeffs[0] = np.arange(d, dtype=np.uint64)
    else:
        coeffs = o.felt_array(SEED, 0, 3 * d).reshape(d, 3).T.copy()
    F = sb.BaseField.main()
    fri = sb.Fri(BFI.generator(), BFI.primitive_nth_root(rec["N"]), rec["N"], rec["expansion"], rec["num_colinearity_tests"], XF)
    t0 = time.time()
    cw = fri.domain.xevaluate(sb.XArray.from_numpy(coeffs), as_array=True)
    soa = cw.to_numpy()
    for i in rec.get("disturb", []):
        soa[:, i] = 0
    cw = sb.XArray.from_numpy(soa)
    sha = hashlib.sha256(np.ascontiguousarray(soa).tobytes()).hexdigest()
    ps = sb.ProofStream()
    if rec["num_prepushed"]:
        r = [hashlib.blake2b(bytes([i])).digest() for i in range(2)]
        e = [XF.from_limbs([o.felt(SEED + 88, 3 * i + k) for k in range(3)]) for i in range(3)]
        for ob in [r[0], (e[0], e[1], e[2]), [r[1]]]:
            ps.push(ob)
    root0 = sb.Merkle(cw).root()
    idx = fri.prove(cw, ps)
    ser = ps.serialize()
    gold = open(os.path.join(G, "fri_%s_stream.bin" % tag), "rb").read()
    vs = sb.ProofStream(); vs.objects = list(ps.objects); vs.read_index = rec["num_prepushed"]
    verdict = fri.verify(vs, root0)
    print(tag, "cw", sha == rec["codeword_sha"], "root0", root0.hex() == rec["roots"][0], "idx", idx == rec["indices"],
          "nobj", len(ps.objects) == rec["num_objects"], "stream", ser == gold, "verify", verdict == rec["verify"], "%.2fs" % (time.time() - t0), flush=True)
