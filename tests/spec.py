"""JSON feature spec (as stored in tests/golden/model_*.npz) -> deepctr_amd feature columns."""
from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat


def _sparse(d):
    return SparseFeat(d["name"], d["vocabulary_size"], d["embedding_dim"], use_hash=d.get("use_hash", False),
                      vocabulary_path=d.get("vocabulary_path"), dtype=d.get("dtype", "int32"),
                      embedding_name=d.get("embedding_name"), group_name=d.get("group_name", "default_group"))


def columns_from_spec(spec):
    cols = []
    for d in spec:
        if d["type"] == "sparse":
            cols.append(_sparse(d))
        elif d["type"] == "dense":
            cols.append(DenseFeat(d["name"], d.get("dimension", 1)))
        else:
            cols.append(VarLenSparseFeat(_sparse(d["sparsefeat"]), d["maxlen"], d.get("combiner", "mean"),
                                         d.get("length_name"), d.get("weight_name"), d.get("weight_norm", True)))
    return cols
