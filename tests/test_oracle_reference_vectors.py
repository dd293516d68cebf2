"""Pins the oracle against the reference's own known-answer tests.

Each test mirrors one gtest case of the reference's test/ndt_model_tests.cpp
(cited per test), with the reference's own tolerances, plus the tighter digits
from SURVEY.md section 4 (hand-emulated reference operation order).
"""
import math

import numpy as np
import pytest

import oracle_lib as O


def test_ndt_cell():
    # reference test/ndt_model_tests.cpp:32-105 (test_ndt_cell)
    cell = O.Cell()
    cell.addPoint(3.5, 3.5)
    cell.addPoint(3.5, 3.5)
    cell.addPoint(3.4, 3.45)
    cell.addPoint(3.6, 3.55)
    assert not cell.valid
    cell.compute()
    assert cell.valid
    assert cell.mean[0] == 3.5  # EXPECT_DOUBLE_EQ
    assert cell.mean[1] == 3.5
    # not enough points yet (:57-59)
    assert cell.score(3.5, 3.5) == pytest.approx(0.0, abs=0.001)
    assert cell.score(3.5, 3.5) == 0.0

    cell.addPoint(3.6, 3.45)
    cell.addPoint(3.4, 3.55)
    cell.compute()
    cov = cell.covariance
    assert cov[0, 0] == pytest.approx(0.008, abs=0.001)
    assert cov[0, 1] == pytest.approx(0.0, abs=0.001)
    assert cov[1, 1] == pytest.approx(0.002, abs=0.001)
    # SURVEY section 4 digits (catastrophic-cancellation noise reproduced)
    assert cov[0, 0] == pytest.approx(0.008000000000001251, rel=1e-12)
    assert cov[1, 1] == pytest.approx(0.0020000000000003127, rel=1e-12)

    assert cell.score(3.5, 3.5) == pytest.approx(1.0, abs=0.001)
    assert cell.score(3.5 + math.sqrt(0.008), 3.5) == pytest.approx(0.6065, abs=0.001)
    assert cell.score(3.5 + 2 * math.sqrt(0.008), 3.5) == pytest.approx(0.1353, abs=0.001)
    assert cell.score(3.5, 3.5 + math.sqrt(0.002)) == pytest.approx(0.6065, abs=0.001)
    assert cell.score(3.5, 3.5 + 2 * math.sqrt(0.002)) == pytest.approx(0.1353, abs=0.001)
    assert cell.score(0.0, 0.0) == pytest.approx(0.0, abs=0.001)
    # SURVEY section 4 digits
    assert cell.score(3.5 + math.sqrt(0.008), 3.5) == pytest.approx(0.60653065971268094, rel=1e-10)
    assert cell.score(3.5 + 2 * math.sqrt(0.008), 3.5) == pytest.approx(0.13533528323665509, rel=1e-9)
    assert cell.score(3.5, 3.5 + math.sqrt(0.002)) == pytest.approx(0.60653065971267794, rel=1e-10)


def test_ndt_cell_no_x_variation():
    # reference test/ndt_model_tests.cpp:107-147
    cell = O.Cell()
    cell.addPoint(3.5, 3.5)
    corr = cell.correlation
    assert corr[0, 0] == 12.25
    assert corr[0, 1] == 12.25
    assert corr[1, 0] == 0.0
    assert corr[1, 1] == 12.25
    cell.addPoint(3.5, 3.45)
    cell.addPoint(3.5, 3.45)
    cell.addPoint(3.5, 3.55)
    cell.addPoint(3.5, 3.55)
    corr = cell.correlation
    # EXPECT_DOUBLE_EQ is 4-ulp equality
    assert corr[0, 0] == pytest.approx(12.25, rel=4 * np.finfo(float).eps)
    assert corr[0, 1] == pytest.approx(12.25, rel=4 * np.finfo(float).eps)
    assert corr[1, 0] == 0.0
    assert corr[1, 1] == pytest.approx(12.252, rel=4 * np.finfo(float).eps)
    cell.compute()
    assert cell.mean[0] == pytest.approx(3.5, rel=4 * np.finfo(float).eps)
    assert cell.mean[1] == pytest.approx(3.5, rel=4 * np.finfo(float).eps)
    cov, info = cell.covariance, cell.information
    assert cov[0, 0] == 0.0
    assert cov[1, 0] == 0.0
    assert cov[0, 1] == 0.0
    assert cov[1, 1] == pytest.approx(0.0025, abs=0.000001)
    assert info[0, 0] == pytest.approx(400000.0, abs=0.000001)
    assert info[1, 0] == 0.0
    assert info[0, 1] == 0.0
    assert info[1, 1] == 0.0
    # SURVEY section 4 digits for the clamp branch
    assert info[0, 0] == pytest.approx(399999.9999998664, rel=1e-12)


def test_ndt_cell_no_y_variation():
    # reference test/ndt_model_tests.cpp:149-189
    cell = O.Cell()
    cell.addPoint(3.5, 3.5)
    corr = cell.correlation
    assert corr[0, 0] == 12.25 and corr[0, 1] == 12.25 and corr[1, 0] == 0.0 and corr[1, 1] == 12.25
    cell.addPoint(3.45, 3.5)
    cell.addPoint(3.45, 3.5)
    cell.addPoint(3.55, 3.5)
    cell.addPoint(3.55, 3.5)
    corr = cell.correlation
    assert corr[0, 0] == pytest.approx(12.252, rel=4 * np.finfo(float).eps)
    assert corr[0, 1] == pytest.approx(12.25, rel=4 * np.finfo(float).eps)
    assert corr[1, 0] == 0.0
    assert corr[1, 1] == pytest.approx(12.25, rel=4 * np.finfo(float).eps)
    cell.compute()
    assert cell.mean[0] == pytest.approx(3.5, rel=4 * np.finfo(float).eps)
    assert cell.mean[1] == pytest.approx(3.5, rel=4 * np.finfo(float).eps)
    cov, info = cell.covariance, cell.information
    assert cov[0, 0] == pytest.approx(0.0025, abs=0.000001)
    assert cov[1, 0] == 0.0
    assert cov[0, 1] == 0.0
    assert cov[1, 1] == 0.0
    assert info[0, 0] == 0.0
    assert info[1, 0] == 0.0
    assert info[0, 1] == 0.0
    assert info[1, 1] == pytest.approx(400000.0, abs=0.000001)


def test_ndt():
    # reference test/ndt_model_tests.cpp:191-230
    ndt = O.NDT(1.0, 10.0, 10.0, -5.0, -5.0)
    points = [(3.5, 3.5), (3.45, 3.4), (3.55, 3.6), (3.45, 3.6), (3.45, 3.6)]
    ndt.addScan((0.0, 0.0, 0.0), points)
    ndt.compute()
    score = ndt.likelihood([(3.5, 3.5)])
    assert score == pytest.approx(0.7659, abs=0.001)
    # SURVEY section 4 digits
    assert ndt.size_x == 11 and ndt.size_y == 11
    assert ndt.getIndex(3.5, 3.5) == 96
    cell = ndt.cell(96)
    assert cell.mean[0] == pytest.approx(3.4799999999999995, rel=1e-15)
    assert cell.mean[1] == pytest.approx(3.54, rel=1e-15)
    assert cell.covariance[0] == pytest.approx(0.002000000000006441, rel=1e-9)
    assert cell.covariance[1] == pytest.approx(0.0010000000000021103, rel=1e-9)
    assert cell.covariance[3] == pytest.approx(0.007999999999999119, rel=1e-9)
    assert cell.information[0] == pytest.approx(533.3333333316551, rel=1e-9)
    assert cell.information[1] == pytest.approx(-66.66666666660493, rel=1e-9)
    assert cell.information[3] == pytest.approx(133.33333333335787, rel=1e-9)
    assert score == pytest.approx(0.76592833836492369, rel=1e-12)


def test_get_index_edges():
    # reference src/ndt_model.cpp:203-218
    ndt = O.NDT(1.0, 10.0, 10.0, -5.0, -5.0)
    assert ndt.getIndex(-5.0, -5.0) == 0  # on the origin: inside
    assert ndt.getIndex(np.nextafter(-5.0, -10.0), 0.0) == -1
    assert ndt.getIndex(0.0, np.nextafter(-5.0, -10.0)) == -1
    # size_x_ = 10/1 + 1 = 11 cells: x in [5, 6) is still cell 10
    assert ndt.getIndex(5.5, -5.0) == 10
    assert ndt.getIndex(6.0, -5.0) == -1
    assert ndt.getIndex(-5.0, 5.999) == 10 * 11
    assert ndt.getIndex(-5.0, 6.0) == -1


def test_search_offsets_counts():
    # SURVEY table T1: the FP-accumulated loops give 21, not 20, for the defaults
    assert len(O.search_offsets(0.05, 0.005)) == 21
    assert len(O.search_offsets(0.1, 0.0025)) == 80
    assert len(O.search_offsets(0.5, 0.05)) == 21
    assert len(O.search_offsets(0.2, 0.01)) == 40
    assert len(O.search_offsets(1.0, 0.02)) == 100
    assert len(O.search_offsets(0.5, 0.005)) == 200
    assert len(O.search_offsets(5.0, 0.02)) == 501
    assert len(O.search_offsets(math.pi, 0.005)) == 1257
    lin = O.search_offsets(0.5, 0.05)
    assert lin[0] == -0.5 and lin[10] != 0.0 and abs(lin[10]) < 1e-15


def test_normalize_angle():
    assert O.lib().orc_normalize_angle(0.0) == 0.0
    assert O.lib().orc_normalize_angle(3 * math.pi) == pytest.approx(math.pi, abs=1e-12) or \
        O.lib().orc_normalize_angle(3 * math.pi) == pytest.approx(-math.pi, abs=1e-12)
    assert O.lib().orc_shortest_angular_distance(3.0, -3.0) == pytest.approx(2 * math.pi - 6.0, abs=1e-12)
