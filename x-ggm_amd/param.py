"""Flag namespace with the reference's names and defaults (src/param.py:34-123).  The
reference parses ``sys.argv`` at import time; here ``args`` is a plain namespace that
``parse_args(argv)`` can refresh, so importing the package never touches the command line."""
import argparse


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--bs", dest="batch_size", type=int, default=8)
    p.add_argument("--lr", type=float, default=1e-5)
    p.add_argument("--epochs", type=int, default=4)
    p.add_argument("--seed", type=int, default=9595)
    p.add_argument("--llayers", type=int, default=9)
    p.add_argument("--xlayers", type=int, default=5)
    p.add_argument("--rlayers", type=int, default=5)
    p.add_argument("--gnn", type=str, default="GCN")
    p.add_argument("--num_layer", type=int, default=2)
    p.add_argument("--sigma", type=float, default=1.0)
    p.add_argument("--delta", type=int, default=5)
    p.add_argument("--fromScratch", dest="from_scratch", action="store_const", default=False, const=True)
    p.add_argument("--multiGPU", action="store_const", default=False, const=True)
    p.add_argument("--vocab", dest="vocab_path", type=str, default=None,
                   help="local BERT vocab.txt (the reference downloads it; offline it must be given)")
    return p


def parse_args(argv=None):
    global args
    args = build_parser().parse_args(argv or [])
    return args


args = build_parser().parse_args([])
