"""indicators.kappa — accuracy metrics (mirror of the reference module of the same name).

`kappa` (indicators/kappa.py:10-22) and `aa_oa` (:69-84) keep the reference's definitions exactly, including
its conventions: the confusion matrix is [prediction][target] (mainsolver.py:141), per-class accuracy and AA
skip class 0, and OA keeps the class-0 column in its denominator (:82).  `expo_result` (:87-118) writes the
same cells through openpyxl when it is importable and a JSON file of the same content otherwise.
"""
import json
import os

import numpy as np


def kappa(matrix):
    m = np.asarray(matrix, dtype=np.float64)
    n = np.sum(m)
    sum_po = 0.0
    sum_pe = 0.0
    for i in range(len(m[0])):
        sum_po += m[i][i]
        sum_pe += np.sum(m[i, :]) * np.sum(m[:, i])
    po = sum_po / n
    pe = sum_pe / (n * n)
    return (po - pe) / (1 - pe)


def _aa_oa(matrix):
    m = np.asarray(matrix, dtype=np.float64)
    accuracy = []
    b = np.sum(m, axis=0)
    c = 0
    on_display = []
    with np.errstate(divide='ignore', invalid='ignore'):
        for i in range(1, m.shape[0]):
            a = m[i][i] / b[i]
            c += m[i][i]
            accuracy.append(a)
            on_display.append([b[i], m[i][i], a])
        aa = np.mean(accuracy)
        oa = c / np.sum(b, axis=0)
        k = kappa(m)
    return aa, oa, k, on_display


def aa_oa(matrix):
    aa, oa, k, on_display = _aa_oa(matrix)
    for i, (tot, ok, acc) in enumerate(on_display, start=1):
        print("Category:{}. Overall:{}. Correct:{}. Accuracy:{:.6f}".format(i, tot, ok, acc))
    print("OA:{:.6f} AA:{:.6f} Kappa:{:.6f}".format(oa, aa, k))
    return [aa, oa, k, on_display]


def aa_oa_quiet(matrix):
    aa, oa, k, _ = _aa_oa(matrix)
    return float(aa), float(oa), float(k)


def expo_result(result, cfg, time, group_num):
    """Per-run result block: per-class rows (count, correct, accuracy), OA / AA / KAPPA, train / test time."""
    aa, oa, k, on_display = result
    savepath = cfg['RESULT_excel']
    try:
        from openpyxl import Workbook, load_workbook
    except ImportError:
        path = os.path.splitext(savepath)[0] + '.json'
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[str(group_num)] = {'per_class': [[float(v) for v in row] for row in on_display], 'OA': float(oa),
                                'AA': float(aa), 'KAPPA': float(k), 'train_time': float(time[0]), 'test_time': float(time[1])}
        json.dump(data, open(path, 'w'), indent=1)
        return path
    col = group_num * 8
    if group_num == 0 or not os.path.exists(savepath):
        wb = Workbook()
        sheet = wb.active
    else:
        wb = load_workbook(savepath)
        sheet = wb.active
    for r, name in enumerate(('Category', 'Overall', 'Correct', 'Accuracy'), start=1):
        sheet.cell(r, col + 1, name)
    for i, (tot, ok, acc) in enumerate(on_display, start=1):
        sheet.cell(1, col + 1 + i, i)
        sheet.cell(2, col + 1 + i, float(tot))
        sheet.cell(3, col + 1 + i, float(ok))
        sheet.cell(4, col + 1 + i, float(acc))
    for c, (name, val) in enumerate((('OA', oa), ('AA', aa), ('KAPPA', k), ('Train time(s)', time[0]), ('Test time(s)', time[1]))):
        sheet.cell(6, col + 1 + 2 * c, name)
        sheet.cell(6, col + 2 + 2 * c, float(val))
    wb.save(savepath)
    return savepath
