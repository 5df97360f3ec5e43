"""Host-side logic without a GPU: the C ABI loads and exports what the header
declares, the artefact layout round-trips, the plugin behaves like the
reference's (src/shrekommender_system/tests/test_als_recommender.py:35-84, but
against real objects instead of mocks), and the product path refuses to run
without a device."""

import os
import pickle
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from movie_recommender_system_amd import ALSRecommender, AlternatingLeastSquares, LightFMScorer, _lib, artifacts
from movie_recommender_system_amd.base import ModelNotLoadedException

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_GPU = not torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mirec.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mirec_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 15
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.mirec_version() >= _lib.ABI_VERSION == 102
    assert lib.mirec_factor_width(2, 0) == 32 and lib.mirec_factor_width(100, 0) == 128
    assert lib.mirec_factor_width(17, 1) == 24 and lib.mirec_factor_width(193, 0) == -1
    assert lib.mirec_als_short_cache_rows(128) >= 8 and lib.mirec_als_short_cache_rows(128) % 2 == 0
    assert lib.mirec_als_short_cache_rows(100) == -1


def test_abi_rejects_bad_arguments_without_touching_a_device():
    lib = _lib.load()
    rc = lib.mirec_topk(None, 4, None, 4, 20, None, None, None, None, 0, None, None, None, 0, None)
    assert rc == -1 and "k must be" in _lib.last_error()
    rc = lib.mirec_topk(None, 4, None, 4, 20, None, None, None, None, 500, None, None, None, 0, None)
    assert rc == -4 and "224" in _lib.last_error()
    rc = lib.mirec_gram(None, 10, 50, 0.1, None, None, None, 0, None)
    assert rc != 0
    with pytest.raises(_lib.MirecError):
        _lib.check(rc, "mirec_gram")


@pytest.mark.skipif(not NO_GPU, reason="checks the no-device failure mode")
def test_scoring_width_and_topk_workspace_sizes():
    """Widths for a top-k pass: the smallest supported one, moved up to a multiple of 16 where that puts k <= 24
    on the bf16-plane kernel; the workspace query answers 0 wherever that kernel does not apply."""
    from movie_recommender_system_amd import engine as E, _lib
    lib = _lib.require()
    assert [E.scoring_width(f, 20) for f in (5, 17, 20, 40, 64, 100, 128, 130)] == [16, 32, 32, 48, 64, 128, 128, 160]
    assert E.scoring_width(17, 100) == 24 and E.scoring_width(128, 100) == 128
    hot = lib.mirec_topk_workspace_bytes
    # header, query list, one norm per item tile, bins and items of an in-call sweep order, the three images of an
    # item tile, and per workgroup in flight (4 of 256 queries here, 1024 at most) a flag and 128 rows of a bf16 plane
    assert hot(1000, 27000, 128, 20) == (256 + 4096 + 3584 + 4096 * 4 + 108032 + 844 * 3 * 32 * 128 * 2
                                         + 256 + 4 * 128 * 128 * 2)
    assert hot(10**6, 27000, 128, 20) - hot(10**6 - 256 * 100, 27000, 128, 20) == 4 * 256 * 100  # only the query list
    assert hot(1000, 27000, 128, 25) == 0 and hot(1000, 27000, 160, 20) == 0 and hot(1000, 27000, 24, 20) == 0
    assert hot(0, 27000, 128, 20) == 0 and hot(1000, 0, 128, 20) == 0


def test_numeric_calls_fail_loudly_without_a_gpu():
    M = sp.csr_matrix(np.array([[1, 0, 1], [0, 1, 0]], dtype=np.float32))
    model = AlternatingLeastSquares(factors=2, iterations=1, random_state=42)
    with pytest.raises(_lib.MirecError, match="no HIP device"):
        model.fit(M)
    sc = LightFMScorer(np.zeros((3, 2)), np.zeros(3), np.zeros((4, 2)), np.zeros(4))
    with pytest.raises(_lib.MirecError):
        sc.predict(np.array([0]), np.array([1]))


def test_build_matrix_matches_reference_semantics():
    """als/als.py:107-123: ids as strings in lexicographic order, duplicates collapse, values 1.0 fp32."""
    users = [10, 9, 10, "a", 9, 10]
    movies = ["m2", "m1", "m2", "m1", "m10", "m1"]
    M, u2i, m2i, i2m = artifacts.build_matrix(users, movies)
    assert list(u2i) == ["10", "9", "a"] and list(m2i) == ["m1", "m10", "m2"]
    assert M.dtype == np.float32 and M.shape == (3, 3) and M.nnz == 5
    assert M.toarray().tolist() == [[1, 0, 1], [1, 1, 0], [1, 0, 0]]
    assert i2m[2] == "m2"


def test_popularity_conventions():
    M = sp.csr_matrix(np.array([[1, 1, 0, 1], [1, 0, 0, 1], [1, 0, 0, 0]], dtype=np.float32))
    a = artifacts.popularity_order(M, "sort_then_reverse")     # als_recommender.py:138-139
    b = artifacts.popularity_order(M, "negate_then_sort")      # als/als.py:175-176
    sums = np.asarray(M.sum(axis=0)).ravel()
    assert np.array_equal(a, np.argsort(sums)[::-1]) and np.array_equal(b, np.argsort(-sums))
    assert a[0] == 0 and b[0] == 0 and set(a[-2:].tolist()) == {1, 2}


@pytest.fixture
def dummy_artefacts():
    train_matrix = sp.csr_matrix(np.array([[1, 0, 1], [0, 1, 0]], dtype=np.float32))
    return {
        "train_matrix": train_matrix,
        "user_mappings": {"user_to_idx": {"u1": 0, "u2": 1}, "n_users": 2},
        "movie_mappings": {"idx_to_movie": {0: "m1", 1: "m2", 2: "m3"}, "movie_to_idx": {"m1": 0, "m2": 1, "m3": 2},
                           "n_movies": 3},
    }


def loaded_plugin(dummy_artefacts):
    """A plugin in the state `train` leaves it in, with fixed factors instead of a GPU fit."""
    rec = ALSRecommender({"artefacts": {}, "metadata": {}}, "als")
    rec.train_matrix = dummy_artefacts["train_matrix"]
    rec.user_mappings = dummy_artefacts["user_mappings"]
    rec.movie_mappings = dummy_artefacts["movie_mappings"]
    model = AlternatingLeastSquares(factors=2, iterations=1, alpha=40, random_state=42)
    model.user_factors = np.array([[1.0, 0.0], [0.0, 1.0]], dtype=np.float32)
    model.item_factors = np.array([[1.0, 0.0], [0.5, 0.5], [0.0, 1.0]], dtype=np.float32)
    rec.model = model
    rec.most_popular_movies = artifacts.popularity_order(rec.train_matrix)
    rec.hyperparameters = {"factors": 2, "iterations": 1}
    rec.is_loaded = True
    return rec


def test_plugin_not_loaded(dummy_artefacts):
    rec = ALSRecommender({"artefacts": dummy_artefacts, "hyperparameters": {"factors": 2, "iterations": 1}})
    with pytest.raises(ModelNotLoadedException):
        rec.recommend_for_user("u1")
    with pytest.raises(ModelNotLoadedException):
        rec.get_model_stats()
    assert rec.health_check()["status"] == "unhealthy"
    with pytest.raises(ValueError, match="Missing required artefact"):
        rec.train({"artefacts": {"train_matrix": dummy_artefacts["train_matrix"]}})


def test_plugin_new_user_profile_stats_and_fallback(dummy_artefacts):
    rec = loaded_plugin(dummy_artefacts)
    new = rec.recommend_for_user("new_user", N=2)
    assert new.user_type == "new" and new.method == "most_popular" and new.total_count == 2
    assert [r["rank"] for r in new.recommendations] == [1, 2] and new.recommendations[0]["score"] is None
    assert new.to_dict()["model_name"] == "als"
    prof = rec.get_user_profile("u1")
    assert prof["user_type"] == "known" and prof["interactions_count"] == 2 and prof["liked_movies"] == ["m1", "m3"]
    assert rec.get_user_profile("nobody")["interactions_count"] == 0
    stats = rec.get_model_stats()
    assert stats["n_users"] == 2 and stats["n_movies"] == 3 and stats["matrix_shape"] == (2, 3)
    assert stats["n_interactions"] == 3 and stats["sparsity"] == pytest.approx(50.0)
    assert rec.health_check()["status"] == "healthy"
    unknown = rec.get_similar_items("mX")
    assert unknown.error is not None and unknown.total_count == 0
    if NO_GPU:
        # scoring raises (no device) -> the reference's contract: log and serve the popular list
        known = rec.recommend_for_user("u1", N=2)
        assert known.user_type == "new" and known.total_count == 2
        sim = rec.get_similar_items("m1", N=2)
        assert sim.error is not None and sim.similar_items == []


def test_save_load_roundtrip_and_artifact_layout(tmp_path, dummy_artefacts):
    rec = loaded_plugin(dummy_artefacts)
    rec.save_model(str(tmp_path / "only_model"))
    blob = pickle.load(open(tmp_path / "only_model" / "als_model.pkl", "rb"))
    assert set(blob) >= {"model", "most_popular_movies", "model_params"}
    # the full models/als@v1 layout
    root = artifacts.write_artifacts(tmp_path / "als@v1", rec.train_matrix, rec.user_mappings["user_to_idx"],
                                     rec.movie_mappings["movie_to_idx"], rec.model,
                                     metadata={"trained_at": "2026-01-01T00:00:00Z"})
    assert sorted(os.listdir(root)) == ["als_model.pkl", "dataset_stats.json", "manifest.yaml", "movie_mappings.pkl",
                                        "train_matrix.npz", "user_mappings.pkl"]
    import yaml
    manifest = yaml.safe_load(open(root / "manifest.yaml"))
    assert {"name", "version", "class_path", "runtime", "artefacts", "metadata"} <= set(manifest)
    assert manifest["class_path"] == "movie_recommender_system_amd.recommender:ALSRecommender"
    # resolve the class_path the way the reference's registry does (serving/model_registry.py:109-118)
    module_name, cls_name = manifest["class_path"].split(":")
    cls = getattr(__import__(module_name, fromlist=[cls_name]), cls_name)
    config = {**manifest["runtime"], "artefacts": manifest["artefacts"], "metadata": manifest["metadata"]}
    fresh = cls(config, "als@v1")
    fresh.load_model(str(root))
    assert fresh.is_loaded and fresh.user_mappings["n_users"] == 2 and fresh.movie_mappings["n_movies"] == 3
    assert np.array_equal(fresh.model.item_factors, rec.model.item_factors)
    assert fresh.model.factors == 2 and fresh.model.alpha == 40.0
    assert np.array_equal(fresh.most_popular_movies, rec.most_popular_movies)
    assert fresh.recommend_for_user("stranger", 3).total_count == 3
    # both mapping styles are present (als/als.py:149-153 and als_recommender.py:85-86)
    um = pickle.load(open(root / "user_mappings.pkl", "rb"))
    assert {"user_to_idx", "idx_to_user", "n_users"} <= set(um)
    state = pickle.load(open(root / "als_model.pkl", "rb"))["model_state"]
    assert state["factors"] == 2 and state["user_factors"].shape == (2, 2)
    fresh.teardown()
    assert fresh.model is None and not fresh.is_loaded
    with pytest.raises(FileNotFoundError):
        cls(config, "x").load_model(str(tmp_path / "missing"))


def test_model_state_is_plain_numpy():
    model = AlternatingLeastSquares(factors=3, regularization=0.5, alpha=2.0, iterations=7, random_state=9)
    model.user_factors = np.ones((2, 3), dtype=np.float32)
    model.item_factors = np.zeros((4, 3), dtype=np.float32)
    clone = pickle.loads(pickle.dumps(model))
    assert clone.factors == 3 and clone.regularization == 0.5 and clone.iterations == 7 and clone.random_state == 9
    assert np.array_equal(clone.user_factors, model.user_factors)
    assert clone.item_norms.tolist() == [1e-10] * 4 or np.allclose(clone.item_norms, 1e-10)
    # the exact solver is an option of the model, carried through the pickle like the others
    exact = pickle.loads(pickle.dumps(AlternatingLeastSquares(use_cg=False)))
    assert exact.use_cg is False and clone.use_cg is True


def test_lightfm_request_path_signature_and_cold_start_draw():
    """``compute_recommendations`` keeps the reference's parameters (hybrid_docker/inference.py:129-134) and
    an unknown user's tokens come from the reference's vocabulary (:168-183), one of each kind, in the
    reference's order of draws; the row builder follows build_coldstart_user_row (:66-83)."""
    import inspect
    import random
    from movie_recommender_system_amd import lightfm as L
    params = list(inspect.signature(L.compute_recommendations).parameters.values())
    assert [p.name for p in params[:4]] == ["user_id", "state", "top_k", "num_threads"]
    assert params[2].default == 20 and params[3].default == 4
    assert (len(L.COLD_START_AGES), len(L.COLD_START_JOBS), len(L.COLD_START_GENDERS)) == (7, 21, 2)
    random.seed(5)
    drawn = L.draw_cold_start_tokens()
    r = random.Random(5)
    assert drawn == [f"age:{r.choice(L.COLD_START_AGES)}", f"job:{r.choice(L.COLD_START_JOBS)}",
                     f"gender:{r.choice(L.COLD_START_GENDERS)}"]
    fmap = {"u0": 0, "u1": 1, "age:65+": 2, "job:farmer": 3, "gender:F": 4}
    row = L.build_coldstart_user_row(fmap, ["age:65+", "job:nothing-like-it", "gender:F"])
    assert row.shape == (1, 5) and row.indices.tolist() == [2, 4] and row.data.tolist() == [1.0, 1.0]
    assert L.build_coldstart_user_row(fmap, ["x"]).nnz == 0
    from types import SimpleNamespace
    assert L._user_feature_mapping({"dataset": SimpleNamespace(_user_feature_mapping=fmap)}) is fmap
    assert L._user_feature_mapping({"user_feature_mapping": {"a": 1}, "dataset": None}) == {"a": 1}


def _golden_interface():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "plugin_interface.json")))


def test_plugin_interface_equals_the_reference_golden():
    """tests/golden/plugin_interface.json was written from the reference's own recommenders/base.py (loaded by
    path in the build container, tests/golden/make_plugin_interface.py): abstract methods, signatures, result
    dataclasses, to_dict outputs, health_check dictionaries, exceptions, optional hooks.  The interface module
    used outside the reference's environment and the plugin class are held to it."""
    import dataclasses
    import inspect
    from movie_recommender_system_amd import base as B
    gold = _golden_interface()
    assert not B.HAVE_REFERENCE_BASE
    Base = B.BaseRecommender
    assert sorted(Base.__abstractmethods__) == gold["abstract_methods"]
    mine = {n: str(inspect.signature(f)) for n, f in inspect.getmembers(Base, inspect.isfunction)
            if not n.startswith("_") or n == "__init__"}
    assert mine == gold["method_signatures"]
    for name, fields in gold["dataclasses"].items():
        got = [{"name": f.name, "type": str(f.type),
                "default": None if f.default is dataclasses.MISSING else repr(f.default),
                "has_default": f.default is not dataclasses.MISSING} for f in dataclasses.fields(getattr(B, name))]
        assert got == fields, name
    rec = B.RecommendationResult("u1", "known", "ALS", [{"movie_id": "m2", "score": 0.5, "rank": 1}], 1, "als@v1")
    ok = B.SimilarItemsResult("m1", "movie", [{"movie_id": "m3", "similarity_score": 0.25, "rank": 1}], 1, "als@v1")
    err = B.SimilarItemsResult("mX", "movie", [], 0, "als@v1", error="Movie not found in training data")
    assert {"recommendation": rec.to_dict(), "similar_ok": ok.to_dict(), "similar_error": err.to_dict()} == gold["to_dict"]
    for name, bases in gold["exceptions"].items():
        assert [b.__name__ for b in getattr(B, name).__mro__[1:-1]] == bases

    class Stub(Base):
        stats = {"n_users": 2}

        def load_model(self, model_dir):
            self.is_loaded = True

        def recommend_for_user(self, user_id, N=10):
            return None

        def get_similar_items(self, item_id, N=10):
            return None

        def get_user_profile(self, user_id):
            return {}

        def get_model_stats(self):
            if isinstance(self.stats, Exception):
                raise self.stats
            return self.stats

    s = Stub({"k": 1}, "stub@v1")
    assert {"config": s.config, "model_name": s.model_name, "logger_suffix": s.logger.name.rsplit(".", 1)[-1]} == \
        gold["constructor_state"]
    health = {"not_loaded": s.health_check()}
    s.load_model("x")
    health["healthy"] = s.health_check()
    s.stats = {}
    health["empty_stats"] = s.health_check()
    s.stats = RuntimeError("boom")
    health["stats_raise"] = s.health_check()
    assert health == gold["health_check"]
    hooks = {"warmup": s.warmup(), "teardown": s.teardown()}
    for name, call in (("save_model", lambda: s.save_model("d")), ("train", lambda: s.train({}))):
        with pytest.raises(NotImplementedError) as info:
            call()
        hooks[name] = {"raises": "NotImplementedError", "message": str(info.value)}
    assert hooks == gold["optional_hooks"]
    with pytest.raises(TypeError):
        Base({}, "x")
    # the plugin: concrete, same public methods with the same parameters as the interface declares
    assert not inspect.isabstract(ALSRecommender)
    for name, sig in gold["method_signatures"].items():
        if name in ("__init__", "train"):
            continue            # the plugin gives model_name a default and train a named parameter, as the reference's does
        assert str(inspect.signature(getattr(ALSRecommender, name))) == sig, name
    assert str(inspect.signature(ALSRecommender.__init__)) == "(self, config: 'Dict[str, Any]', model_name: 'str' = 'als')"
    assert str(inspect.signature(ALSRecommender.train)) == "(self, training_config: 'Dict[str, Any]') -> 'None'"


@pytest.mark.skipif(not NO_GPU, reason="checks the no-device failure mode of warmup")
def test_registry_sequence_fails_loudly_at_warmup_without_a_device(tmp_path, dummy_artefacts):
    """serving/model_registry.py:48-73: cls(config, model_name=id) -> load_model -> warmup.  Loading is I/O and
    works anywhere; warmup touches the device and must raise here rather than leave a model that answers every
    request from the popular list."""
    rec = loaded_plugin(dummy_artefacts)
    root = artifacts.write_artifacts(tmp_path / "als@v1", rec.train_matrix, rec.user_mappings["user_to_idx"],
                                     rec.movie_mappings["movie_to_idx"], rec.model)
    import yaml
    manifest = yaml.safe_load(open(root / "manifest.yaml"))
    module_name, _, cls_name = manifest["class_path"].partition(":")
    cls = getattr(__import__(module_name, fromlist=[cls_name]), cls_name)
    config = dict(manifest["runtime"])
    config.setdefault("artefacts", manifest["artefacts"])
    config.setdefault("metadata", manifest["metadata"])
    inst = cls(config, model_name="als@v1")
    inst.load_model(str(root))
    assert inst.is_loaded
    with pytest.raises(_lib.MirecError, match="no HIP device"):
        inst.warmup()


REFERENCE_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE_SRC, "shrekommender_system")),
                    reason="needs the reference tree (present in the build container only)")
def test_inside_the_reference_environment():
    """With the reference's package importable -- `implicit` resolved by this package's stand-in
    (movie_recommender_system_amd/compat), as a deployment that has no implicit wheel would do it -- the plugin
    subclasses the REFERENCE's BaseRecommender (what serving/model_registry.py:116-117 checks), the registry's
    own `_resolve_class` accepts the manifest's class_path, the reference's unmodified ALSRecommender builds
    this package's model, and the reference's own test file for the plugin passes.  Runs in child processes:
    this process has already imported the interface without the reference."""
    import subprocess
    import sys
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "movie_recommender_system_amd", "compat"), REFERENCE_SRC, ROOT]))
    code = """
import implicit, implicit.als
from shrekommender_system.recommenders import base as RB
from shrekommender_system.recommenders.als_recommender import ALSRecommender as RefPlugin
from movie_recommender_system_amd import base as B, ALSRecommender, AlternatingLeastSquares
assert B.HAVE_REFERENCE_BASE and B.BaseRecommender is RB.BaseRecommender
assert issubclass(ALSRecommender, RB.BaseRecommender)
assert B.RecommendationResult is RB.RecommendationResult and B.ModelNotLoadedException is RB.ModelNotLoadedException
assert implicit.als.AlternatingLeastSquares is AlternatingLeastSquares
import importlib
module_name, _, class_name = "movie_recommender_system_amd.recommender:ALSRecommender".partition(":")
cls = getattr(importlib.import_module(module_name), class_name)
assert issubclass(cls, RB.BaseRecommender)          # serving/model_registry.py:109-118, step for step
inst = cls({"use_gpu": True, "artefacts": {}, "metadata": {}}, model_name="als@v1")
assert inst.health_check() == {"model_name": "als@v1", "status": "unhealthy", "reason": "Model not loaded"}
import numpy as np, scipy.sparse as sp
ref = RefPlugin({"artefacts": {}, "hyperparameters": {"factors": 2, "iterations": 1}})
ref.train_matrix = sp.csr_matrix([[1, 0, 1], [0, 1, 0]])
try:
    ref._train_model({"factors": 2, "iterations": 1})
    raise SystemExit("a fit without a device must fail")
except Exception as exc:
    assert type(exc).__name__ == "MirecError" and "no HIP device" in str(exc), exc
m = ref.model
assert isinstance(m, AlternatingLeastSquares) and (m.factors, m.iterations, m.regularization, m.alpha, m.random_state) == (2, 1, 0.01, 40.0, 42)
print("ok")
"""
    if NO_GPU:
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0 and res.stdout.strip().endswith("ok"), res.stdout + res.stderr
    ref_test = os.path.join(REFERENCE_SRC, "shrekommender_system", "tests", "test_als_recommender.py")
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", ref_test], env=env,
                         capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    # Five of its six tests pass.  The sixth, test_save_model (:70-78), pickles the autospec'd mock that stands for
    # the model, which unittest.mock cannot do for ANY class behind the mock (pickle rejects the mock's borrowed
    # __class__: "args[0] from __newobj__ args has the wrong class"), the real implicit class included; with a
    # real model object instead of the mock the same save / load is exercised by this suite's own tests.
    assert "5 passed" in res.stdout and "1 failed" in res.stdout, res.stdout + res.stderr
    assert "test_save_model" in res.stdout and "PicklingError" in res.stdout, res.stdout + res.stderr


def test_optional_arguments_are_checked_before_any_device_call():
    """``items`` together with ``filter_items`` is refused as implicit refuses it, an item outside the catalogue
    too -- on the host, before the model asks for a device."""
    model = AlternatingLeastSquares(factors=4)
    model.user_factors = np.ones((3, 4), dtype=np.float32)
    model.item_factors = np.ones((5, 4), dtype=np.float32)
    row = sp.csr_matrix(np.array([[1, 0, 0, 1, 0]], dtype=np.float32))
    with pytest.raises(ValueError, match="both filter_items and items"):
        model.recommend(0, row, N=2, items=[1, 2], filter_items=[3])
    with pytest.raises(ValueError, match="both filter_items and items"):
        model.similar_items(0, N=2, items=[1, 2], filter_items=[3])
    with pytest.raises(IndexError):
        model.recommend(0, row, N=2, items=[1, 5])
    with pytest.raises(ValueError, match="recalculate_user needs user_items"):
        model.recommend(0, None, N=2, recalculate_user=True)
