"""Metric registry (mirror of /root/reference/modeling_utils/modeling_utils/metrics/__init__.py)."""
import typing as tp

import pydantic

from .base import (BaseMetricConfig, GroupedMetric, GroupedMetricConfig, MultidimPearsonCorrCoef,  # noqa: F401
                   MultidimPearsonCorrCoefConfig, OnlinePearsonCorr)

MetricConfig = tp.Annotated[tp.Union[MultidimPearsonCorrCoefConfig, GroupedMetricConfig],
                            pydantic.Field(discriminator="name")]
