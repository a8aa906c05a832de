"""Golden-vector tooling: executes the reference's OWN class bodies on CPU.

Runs only in the build container (reads /root/reference at run time, never copies it;
nothing under tools/ is imported by the product, the tests or the bench).

Recipe: parse the reference file as text with ``ast``, keep only the
``ClassDef`` nodes and ``exec`` them in a namespace that supplies the module globals the
reference reads (its import-time side effects -- hard-coded CSV path, ``.to('cuda')``,
missing ``models/dcrnn.py`` -- are thereby never run).  For the shape-generic
``(B,C,N,T)`` configs the two ``view`` assignments of ``gwnet.forward``
(graph_wavenet.py:189 and :255) are dropped from the AST.
"""
import ast
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"


def _classes(path, drop_lines=(), rewrite_cuda=False):
    src = open(path).read()
    tree = ast.parse(src)
    classes = [n for n in tree.body if isinstance(n, ast.ClassDef)]

    class Drop(ast.NodeTransformer):
        def visit_Assign(self, node):
            if node.lineno in drop_lines:
                return None
            return self.generic_visit(node)

        def visit_Constant(self, node):
            if rewrite_cuda and node.value == "cuda":
                return ast.copy_location(ast.Constant("cpu"), node)
            return node

    mod = ast.Module(body=[Drop().visit(c) for c in classes], type_ignores=[])
    ast.fix_missing_locations(mod)
    return mod


def load_gwnet(generic, default_supports, n_counties=67, feature_vector_size=256,
               time_embed_size=64):
    """Return the reference ``gwnet`` class (plus nconv/linear/gcn).

    generic=True drops graph_wavenet.py:189/:255 so (B,C,N,T) inputs pass straight through.
    """
    path = f"{REF}/models/graph_wavenet.py"
    mod = _classes(path, drop_lines=(189, 255) if generic else ())
    ns = dict(torch=torch, nn=nn, F=F, np=np, n_counties=n_counties,
              feature_vector_size=feature_vector_size, time_embed_size=time_embed_size,
              default_supports=default_supports)
    exec(compile(mod, "ref_gwnet", "exec"), ns)
    return ns


def load_unet(gwnet_cls, n_counties=67, image_dimension=128):
    path = f"{REF}/models/unet.py"
    mod = _classes(path, rewrite_cuda=True)
    ns = dict(torch=torch, nn=nn, F=F, np=np, n_counties=n_counties, feature_vector_size=256,
              time_embed_size=64, loc_embed_size=256, compression_factor=4,
              image_dimension=image_dimension, gwnet=gwnet_cls, DCRNNModel=None,
              default_kwargs={})
    exec(compile(mod, "ref_unet", "exec"), ns)
    return ns


def load_date2vec():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import date2vec  # plain import works
    return date2vec


def load_asym_adj():
    """utils.py cannot be imported (torchvision); extract the one pure function we need."""
    src = open(f"{REF}/utils.py").read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "asym_adj"]
    import scipy.sparse as sp
    ns = dict(np=np, sp=sp)
    mod = ast.Module(body=fn, type_ignores=[])
    exec(compile(mod, "ref_utils", "exec"), ns)
    return ns["asym_adj"]


def fill_params(module, seed, scale=None):
    """Overwrite every parameter/buffer deterministically in state_dict order
    (numpy RandomState(seed)); fixtures then only need the seed."""
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if name.endswith("num_batches_tracked"):
                t.zero_()
                continue
            if name.endswith("running_var"):
                v = rs.uniform(0.5, 1.5, size=tuple(t.shape))
            elif name.endswith("running_mean"):
                v = rs.uniform(-0.2, 0.2, size=tuple(t.shape))
            elif t.dim() <= 1:
                if name.endswith("weight"):   # BN gamma
                    v = rs.uniform(0.5, 1.5, size=tuple(t.shape))
                else:
                    v = rs.uniform(-0.3, 0.3, size=tuple(t.shape))
            else:
                fan_in = int(np.prod(t.shape[1:])) if "nodevec" not in name else 10
                bound = 1.0 / np.sqrt(max(fan_in, 1))
                if "nodevec" in name:
                    v = rs.standard_normal(size=tuple(t.shape))
                else:
                    v = rs.uniform(-bound, bound, size=tuple(t.shape)) * 1.7
            t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
