"""Import the upstream reference (read-only at /root/reference) inside THIS container only.

Used by tests/golden/make_golden.py to generate the committed golden vectors. Nothing in
tests/, bench.py or the package imports this module at run time: /root/reference does not
exist on the GPU box.  Three stub modules stand in for dependencies that are absent here and
that the hot path never executes: easydict, mvtracker.datasets(.utils),
mvtracker.utils.visualizer_mp4.
"""
import sys
import types

REF_ROOT = "/root/reference"


def import_reference():
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")
        m.EasyDict = dict
        sys.modules["easydict"] = m
    if "mvtracker.datasets" not in sys.modules:
        pkg = types.ModuleType("mvtracker.datasets")
        pkg.__path__ = []
        sys.modules["mvtracker.datasets"] = pkg
        u = types.ModuleType("mvtracker.datasets.utils")

        def transform_scene(*a, **k):  # only referenced by a dead branch of the reference
            raise NotImplementedError

        u.transform_scene = transform_scene
        sys.modules["mvtracker.datasets.utils"] = u
    if "mvtracker.utils.visualizer_mp4" not in sys.modules:
        v = types.ModuleType("mvtracker.utils.visualizer_mp4")

        class MultiViewVisualizer:  # never instantiated without save_debug_logs
            def __init__(self, *a, **k):
                raise NotImplementedError

        v.MultiViewVisualizer = MultiViewVisualizer
        sys.modules["mvtracker.utils.visualizer_mp4"] = v
    import mvtracker.models.core.mvtracker.mvtracker as ref_mvt
    import mvtracker.models.evaluation_predictor_3dpt as ref_pred
    import mvtracker.models.core.spatracker.blocks as ref_spa
    import mvtracker.models.core.cotracker2.blocks as ref_co
    import mvtracker.models.core.model_utils as ref_mu
    import mvtracker.models.core.embeddings as ref_emb
    return types.SimpleNamespace(mvt=ref_mvt, pred=ref_pred, spa=ref_spa, co=ref_co, mu=ref_mu, emb=ref_emb)
