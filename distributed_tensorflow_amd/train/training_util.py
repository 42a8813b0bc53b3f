"""``global_step`` helpers."""
from __future__ import annotations

from typing import Optional

from ..framework import dtypes
from ..framework.graph import GraphKeys, get_default_graph
from ..framework.variables import Variable, constant_initializer, get_variable


def get_global_step(graph=None) -> Optional[Variable]:
    graph = graph or get_default_graph()
    steps = graph.get_collection(GraphKeys.GLOBAL_STEP)
    if steps:
        return steps[0]
    return graph._var_store.get("global_step")


def create_global_step(graph=None) -> Variable:
    graph = graph or get_default_graph()
    if get_global_step(graph) is not None:
        raise ValueError('"global_step" already exists.')
    with graph.as_default():
        step = get_variable("global_step", [], dtype=dtypes.float32,
                            initializer=constant_initializer(0), trainable=False)
    graph.add_to_collection(GraphKeys.GLOBAL_STEP, step)
    return step


def get_or_create_global_step(graph=None) -> Variable:
    return get_global_step(graph) or create_global_step(graph)
