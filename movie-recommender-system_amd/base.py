"""Plugin interface of the serving layer.

The reference's registry instantiates ``class_path`` and insists on
``issubclass(cls, BaseRecommender)`` with ITS base class
(``serving/model_registry.py:109-118``), so inside the reference's serving
environment the real ``shrekommender_system.recommenders.base`` is imported
and used.  Outside it (tests, benchmarks, the GPU box) an equivalent interface
is defined here with the same names, fields and defaults
(``recommenders/base.py:13-153``).
"""

from __future__ import annotations

import logging
from abc import ABC, abstractmethod
from dataclasses import dataclass, asdict
from typing import Any, Dict, List, Optional

try:  # pragma: no cover - only inside the reference's serving environment
    from shrekommender_system.recommenders.base import (  # type: ignore
        BaseRecommender,
        ItemNotFoundException,
        ModelNotLoadedException,
        RecommendationResult,
        RecommenderException,
        SimilarItemsResult,
        UserNotFoundException,
    )
    HAVE_REFERENCE_BASE = True
except Exception:  # noqa: BLE001 - any import problem means "not in that environment"
    HAVE_REFERENCE_BASE = False

    @dataclass
    class RecommendationResult:
        user_id: str
        user_type: str          # "known" | "new"
        method: str             # "ALS" | "most_popular"
        recommendations: List[Dict[str, Any]]
        total_count: int
        model_name: str

        def to_dict(self) -> Dict[str, Any]:
            return asdict(self)

    @dataclass
    class SimilarItemsResult:
        item_id: str
        item_type: str
        similar_items: List[Dict[str, Any]]
        total_count: int
        model_name: str
        error: Optional[str] = None

        def to_dict(self) -> Dict[str, Any]:
            d = asdict(self)
            if not self.error:
                d.pop("error")
            return d

    class RecommenderException(Exception):
        pass

    class ModelNotLoadedException(RecommenderException):
        pass

    class UserNotFoundException(RecommenderException):
        pass

    class ItemNotFoundException(RecommenderException):
        pass

    class BaseRecommender(ABC):
        def __init__(self, config: Dict[str, Any], model_name: str):
            self.config = config
            self.model_name = model_name
            self.is_loaded = False
            self.logger = logging.getLogger(f"{__name__}.{model_name}")

        @abstractmethod
        def load_model(self, model_dir: str) -> None: ...

        @abstractmethod
        def recommend_for_user(self, user_id: str, N: int = 10) -> RecommendationResult: ...

        @abstractmethod
        def get_similar_items(self, item_id: str, N: int = 10) -> SimilarItemsResult: ...

        @abstractmethod
        def get_user_profile(self, user_id: str) -> Dict[str, Any]: ...

        @abstractmethod
        def get_model_stats(self) -> Dict[str, Any]: ...

        def health_check(self) -> Dict[str, Any]:
            try:
                if not self.is_loaded:
                    return {"model_name": self.model_name, "status": "unhealthy", "reason": "Model not loaded"}
                stats = self.get_model_stats()
                if not stats:
                    return {"model_name": self.model_name, "status": "unhealthy", "reason": "Failed to fetch stats"}
                return {"model_name": self.model_name, "status": "healthy", "stats": stats}
            except Exception as exc:  # noqa: BLE001
                return {"model_name": self.model_name, "status": "unhealthy", "reason": str(exc)}

        def warmup(self) -> None:
            pass

        def teardown(self) -> None:
            pass

        def save_model(self, model_dir: str) -> None:
            raise NotImplementedError(f"Model saving not implemented for {self.model_name}")

        def train(self, *args, **kwargs) -> None:
            raise NotImplementedError(f"Training is not supported for {self.model_name}")
