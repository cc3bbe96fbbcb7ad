#!/usr/bin/env python3
"""verify_h5.py <sig2seq.h5> [--sig-config models/sig2seq.yaml | none] [--by-name out.rdnw]

The check SURVEY.md 8c / H1 says must happen the day a real `sig2seq.h5` appears: print the file's weight tensors -- names, shapes,
in the file's own order (the order Keras' positional load_weights consumes, radian/model.py:42-45) -- beside the graph this backend
implements (model.py:52-89 + keras-tcn 3.5's residual block: per block conv1D_0, conv1D_1 kernel+bias, matching_conv1D in block 0 only;
dense, dense_1), and say for every tensor whether shape and name agree with its position.  Exit code 0 = the file loads as it is
(`--sig-model <file>`), 1 = it does not.

--by-name out.rdnw: place every tensor by what its NAME says instead of by position and write the packed blob `--sig-model out.rdnw`
accepts -- for a file whose within-layer order differs from this backend's assumption but whose names are keras-tcn's.  No GPU needed."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[1])
    ap.add_argument("h5")
    ap.add_argument("--sig-config", default="none", help="sig2seq.yaml (its dilations x nb_stacks define the graph); none = sig2seq.yaml's own values")
    ap.add_argument("--by-name", metavar="OUT.rdnw", default=None)
    args = ap.parse_args(argv)
    from radian_amd import h5weights, weights
    from radian_amd.basecall import load_dilations
    dil = load_dilations(args.sig_config)
    tensors = h5weights.list_keras_tensors(args.h5)
    rows, problems, unverified = h5weights.check_tensor_list(tensors, dil)
    print(f"{args.h5}: {len(tensors)} tensors; graph: {len(dil)} residual blocks, dilations {tuple(dil)}, {weights.n_params(dil)} parameters")
    wn = max([len(r[1] or "") for r in rows] + [10])
    we = max([len(r[3] or "") for r in rows] + [10])
    print(f"{'#':>3}  {'in the file':<{wn}}  {'shape':<15}  {'graph position (load_weights order)':<{we}}  {'shape':<15}  verdict")
    for i, name, shape, ename, eshape, verdict in rows:
        print(f"{i:>3}  {name or '-':<{wn}}  {str(shape or '-'):<15}  {ename or '-':<{we}}  {str(eshape or '-'):<15}  {verdict}")
    for p in problems:
        print("PROBLEM:", p)
    if unverified:
        print(f"NOTE: {len(unverified)} same-shaped conv tensors carry names this reader does not know; their order is taken on trust")
    if args.by_name:
        flat = h5weights.read_keras_weights(args.h5, dil, order="by_name")
        with open(args.by_name, "wb") as f:
            f.write(weights.pack_blob(flat, dil))
        print(f"wrote {args.by_name}: {flat.size} parameters placed by name")
        return 0
    print("OK: loads as it is" if not problems else "FAILED: does not load positionally" + ("" if len(tensors) != len(rows) else "; if the names are right, --by-name converts it"))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
