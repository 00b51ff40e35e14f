"""Mirror of python/repair/tests/test_utils.py: option parsing and the argument type checker (same messages)."""
import re
from typing import Dict, List, Union

import pytest

from repair.utils import argtype_check, get_option_value


class BaseClass:
    def __init__(self, n: int) -> None:
        self.n = n

    def __eq__(self, other):
        return isinstance(other, BaseClass) and self.n == other.n


class DerivedClass(BaseClass):
    pass


@argtype_check
def f_int(v: int) -> int:
    return v


@argtype_check
def f_float(v: float) -> float:
    return v


@argtype_check
def f_str(v: str) -> str:
    return v


@argtype_check
def f_class(v: BaseClass) -> BaseClass:
    return v


@argtype_check
def f_union(v: Union[int, List[str], Dict[str, int]]) -> Union[int, List[str], Dict[str, int]]:
    return v


@argtype_check
def f_list(v: List[str]) -> List[str]:
    return v


@argtype_check
def f_class_list(v: List[BaseClass]) -> List[BaseClass]:
    return v


@argtype_check
def f_dict(v: Dict[str, int]) -> Dict[str, int]:
    return v


def _raises(msg, fn):
    with pytest.raises(TypeError, match=re.escape(msg)):
        fn()


def test_get_option_value():
    options = {"key1": "abcd", "key2": "1", "key3": "3.2"}
    assert get_option_value(options, "key1", "efgh", type_class=str) == "abcd"
    assert get_option_value(options, "key2", 3, type_class=int) == 1
    assert get_option_value(options, "key3", 0.0, type_class=float) == 3.2
    assert get_option_value(options, "key2", False, type_class=bool) is True
    assert get_option_value(options, "non.existent", "efgh", type_class=str) == "efgh"
    assert get_option_value(options, "non.existent", 3, type_class=int) == 3
    assert get_option_value(options, "non.existent", 0.0, type_class=float) == 0.0
    assert get_option_value(options, "non.existent", False, type_class=bool) is False
    for key, default, t, msg in (("key1", 2, int, 'Failed to cast "abcd" into int data: key=key1'),
                                 ("key1", 0.0, float, 'Failed to cast "abcd" into float data: key=key1'),
                                 ("key3", 2, int, 'Failed to cast "3.2" into int data: key=key3')):
        with pytest.raises(ValueError, match=re.escape(msg)):
            get_option_value(options, key, default, type_class=t)


def test_primitive_and_class_type_check():
    _raises("`v` should be provided as int, got str", lambda: f_int("a"))
    _raises("`v` should be provided as float, got int", lambda: f_float(1))
    _raises("`v` should be provided as str, got int", lambda: f_str(1))
    assert f_int(1) == 1 and f_float(2.0) == 2.0 and f_str("a") == "a"
    _raises("`v` should be provided as BaseClass, got int", lambda: f_class(1))
    assert f_class(BaseClass(1)) == BaseClass(1) and f_class(DerivedClass(1)) == DerivedClass(1)


def test_union_list_dict_type_check():
    _raises("`v` should be provided as int/list[str]/dict[str,int], got str", lambda: f_union("a"))
    _raises("`v` should be provided as int/list[str]/dict[str,int], got list", lambda: f_union([1, 2, 3]))
    _raises("`v` should be provided as int/list[str]/dict[str,int], got dict", lambda: f_union({1: 1, 2: 2}))
    assert f_union(1) == 1 and f_union(["a", "b"]) == ["a", "b"] and f_union({"a": 1, "b": 2}) == {"a": 1, "b": 2}
    _raises("`v` should be provided as list[str], got int", lambda: f_list(1))
    _raises("`v` should be provided as list[str], got int in elements", lambda: f_list([1, 2, 3]))
    _raises("`v` should be provided as list[BaseClass], got int", lambda: f_class_list(1))
    _raises("`v` should be provided as list[BaseClass], got int in elements", lambda: f_class_list([1, 2, 3]))
    _raises("`v` should be provided as list[BaseClass], got str in elements", lambda: f_class_list([BaseClass(1), "a"]))
    assert f_list(["a", "b", "c"]) == ["a", "b", "c"]
    assert f_class_list([BaseClass(1), DerivedClass(2)]) == [BaseClass(1), DerivedClass(2)]
    _raises("`v` should be provided as dict[str,int], got str", lambda: f_dict("a"))
    _raises("`v` should be provided as dict[str,int], got list", lambda: f_dict([1, 2, 3]))
    _raises("`v` should be provided as dict[str,int], got int in keys", lambda: f_dict({1: 1, 2: 2}))
    _raises("`v` should be provided as dict[str,int], got str in values", lambda: f_dict({"a": "1", "b": "2"}))
    assert f_dict({"a": 1, "b": 2}) == {"a": 1, "b": 2}
