/* Host-side helper of the drop-in boundary (CPython extension, no GPU code): the three bulk conversions between the
 * reference's Python value objects and flat arrays that dominate a call once the GPU pass itself takes milliseconds —
 * 50 000 matches in, 31 841 inlier pairs out spent 32 ms in CPython object handling around a 2.4 ms pass
 * (profiles/r02/api_c3_profile.log).  Semantics are those of the Python code they replace in epipolar/_engine.py and
 * epipolar_ransac.py (reference lib/epipolar/epipolar_ransac.py:55-57, lib/ransac/ransac.py:59-64,76):
 *
 *   match_pairs(features_a, features_b, matches)  -> [(features_a[m.a_index], features_b[m.b_index]) for m in matches]
 *   pair_arrays(cls, pairs, out)                  -> fills out[2][n][2] (float64) with x, y of both features of each pair
 *   copy_pairs(cls, pairs, order)                 -> [deepcopy(pairs[i]) for i in order] for pairs of exact `cls`
 *                                                    instances whose attribute values are atomic (float / int / bool /
 *                                                    None / str): the copy is a new instance holding the same
 *                                                    immutable values, which is what copy.deepcopy produces for them
 *
 * Each returns None (pair_arrays: False) when the input is not of the plain shape it handles; the caller then runs the
 * general Python path.  Nothing here is needed for correctness.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *s_x, *s_y, *s_a_index, *s_b_index;

/* attribute of a plain instance straight from its __dict__ (borrowed), falling back to getattr (new reference held in *owned) */
static PyObject* plain_attr(PyObject* obj, PyObject* name, PyObject** owned) {
    *owned = NULL;
    PyObject** dictptr = _PyObject_GetDictPtr(obj);
    if (dictptr != NULL && *dictptr != NULL) {
        PyObject* v = PyDict_GetItemWithError(*dictptr, name);
        if (v != NULL) return v;
        if (PyErr_Occurred()) return NULL;
    }
    *owned = PyObject_GetAttr(obj, name);
    return *owned;
}

static PyObject* match_pairs(PyObject* self, PyObject* args) {
    PyObject *fa, *fb, *matches;
    if (!PyArg_ParseTuple(args, "OOO", &fa, &fb, &matches)) return NULL;
    if (!PyList_CheckExact(fa) || !PyList_CheckExact(fb) || !PyList_CheckExact(matches)) Py_RETURN_NONE;
    const Py_ssize_t n = PyList_GET_SIZE(matches), na = PyList_GET_SIZE(fa), nb = PyList_GET_SIZE(fb);
    PyObject* out = PyList_New(n);
    if (out == NULL) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* m = PyList_GET_ITEM(matches, i);
        PyObject *oa, *ob;
        PyObject* ia = plain_attr(m, s_a_index, &oa);
        PyObject* ib = ia ? plain_attr(m, s_b_index, &ob) : NULL;
        if (ia == NULL || ib == NULL) {
            Py_XDECREF(oa);
            Py_DECREF(out);
            return NULL;
        }
        Py_ssize_t a = PyLong_CheckExact(ia) ? PyLong_AsSsize_t(ia) : -1;
        Py_ssize_t b = PyLong_CheckExact(ib) ? PyLong_AsSsize_t(ib) : -1;
        const int exact = PyLong_CheckExact(ia) && PyLong_CheckExact(ib);
        Py_XDECREF(oa);
        Py_XDECREF(ob);
        if (!exact || PyErr_Occurred()) {   /* numpy integers, negative-overflow, ...: the Python path handles them */
            PyErr_Clear();
            Py_DECREF(out);
            Py_RETURN_NONE;
        }
        if (a < 0) a += na;                 /* list indexing semantics */
        if (b < 0) b += nb;
        if (a < 0 || a >= na || b < 0 || b >= nb) {
            Py_DECREF(out);
            PyErr_SetString(PyExc_IndexError, "list index out of range");
            return NULL;
        }
        PyObject* pair = PyTuple_Pack(2, PyList_GET_ITEM(fa, a), PyList_GET_ITEM(fb, b));
        if (pair == NULL) {
            Py_DECREF(out);
            return NULL;
        }
        PyList_SET_ITEM(out, i, pair);
    }
    return out;
}

static int coordinate(PyObject* feature, PyObject* name, double* out) {
    PyObject* owned;
    PyObject* v = plain_attr(feature, name, &owned);
    if (v == NULL) return -1;
    *out = PyFloat_CheckExact(v) ? PyFloat_AS_DOUBLE(v) : PyFloat_AsDouble(v);
    Py_XDECREF(owned);
    return (*out == -1.0 && PyErr_Occurred()) ? -1 : 0;
}

static PyObject* pair_arrays(PyObject* self, PyObject* args) {
    PyObject *cls, *pairs, *out_obj;
    if (!PyArg_ParseTuple(args, "OOO", &cls, &pairs, &out_obj)) return NULL;
    if (!PyList_CheckExact(pairs)) Py_RETURN_FALSE;
    Py_buffer view;
    if (PyObject_GetBuffer(out_obj, &view, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) return NULL;
    const Py_ssize_t n = PyList_GET_SIZE(pairs);
    if (view.len != (Py_ssize_t)(4 * n * sizeof(double))) {
        PyBuffer_Release(&view);
        PyErr_SetString(PyExc_ValueError, "pair_arrays: out must hold 2 x n x 2 float64");
        return NULL;
    }
    double* first = (double*)view.buf;
    double* second = first + 2 * n;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* pair = PyList_GET_ITEM(pairs, i);
        if (!PyTuple_CheckExact(pair) || PyTuple_GET_SIZE(pair) != 2 || (PyObject*)Py_TYPE(PyTuple_GET_ITEM(pair, 0)) != cls ||
            (PyObject*)Py_TYPE(PyTuple_GET_ITEM(pair, 1)) != cls) {
            PyBuffer_Release(&view);
            Py_RETURN_FALSE;
        }
        if (coordinate(PyTuple_GET_ITEM(pair, 0), s_x, first + 2 * i) || coordinate(PyTuple_GET_ITEM(pair, 0), s_y, first + 2 * i + 1) ||
            coordinate(PyTuple_GET_ITEM(pair, 1), s_x, second + 2 * i) || coordinate(PyTuple_GET_ITEM(pair, 1), s_y, second + 2 * i + 1)) {
            PyBuffer_Release(&view);
            return NULL;
        }
    }
    PyBuffer_Release(&view);
    Py_RETURN_TRUE;
}

static int atomic_value(PyObject* v) {   /* copy.deepcopy returns these objects themselves */
    return PyFloat_CheckExact(v) || PyLong_CheckExact(v) || v == Py_None || PyBool_Check(v) || PyUnicode_CheckExact(v);
}

/* new instance of `type` with a copy of src's __dict__; NULL + no error set: not a plain instance */
static PyObject* copy_instance(PyTypeObject* type, PyObject* src, PyObject* empty) {
    PyObject** dictptr = _PyObject_GetDictPtr(src);
    if (dictptr == NULL || *dictptr == NULL) return NULL;
    PyObject *key, *value;
    Py_ssize_t pos = 0;
    while (PyDict_Next(*dictptr, &pos, &key, &value))
        if (!atomic_value(value)) return NULL;
    PyObject* fresh = type->tp_new(type, empty, NULL);
    if (fresh == NULL) return NULL;
    /* PyDict_Copy of an instance's key-sharing __dict__ shares the keys again (a values array per copy, as the
     * constructor makes) and is faster than setting the attributes one by one (5.3 vs 7.5 ms for 31 841 pairs) */
    PyObject* dict = PyDict_Copy(*dictptr);
    if (dict == NULL || PyObject_GenericSetDict(fresh, dict, NULL) != 0) {
        Py_XDECREF(dict);
        Py_DECREF(fresh);
        return NULL;
    }
    Py_DECREF(dict);
    return fresh;
}

static PyObject* copy_pairs(PyObject* self, PyObject* args) {
    PyObject *cls, *pairs, *order_obj;
    if (!PyArg_ParseTuple(args, "OOO", &cls, &pairs, &order_obj)) return NULL;
    if (!PyType_Check(cls) || !PyList_CheckExact(pairs)) Py_RETURN_NONE;
    PyTypeObject* type = (PyTypeObject*)cls;
    /* a plain Python class: instances carry a __dict__, object.__new__ makes them, no __slots__ state to copy */
    if (type->tp_dictoffset == 0 || type->tp_new == NULL || type->tp_itemsize != 0) Py_RETURN_NONE;
    Py_buffer view;
    if (PyObject_GetBuffer(order_obj, &view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return NULL;
    if (view.itemsize != 8 || view.format == NULL || (view.format[0] != 'l' && view.format[0] != 'q')) {
        PyBuffer_Release(&view);
        PyErr_SetString(PyExc_TypeError, "copy_pairs: order must be a contiguous int64 array");
        return NULL;
    }
    const int64_t* order = (const int64_t*)view.buf;
    const Py_ssize_t m = view.len / 8, n = PyList_GET_SIZE(pairs);
    PyObject* empty = PyTuple_New(0);
    PyObject* out = empty ? PyList_New(m) : NULL;
    if (out == NULL) {
        Py_XDECREF(empty);
        PyBuffer_Release(&view);
        return NULL;
    }
    int plain = 1;
    for (Py_ssize_t k = 0; k < m && plain; ++k) {
        const int64_t i = order[k];
        if (i < 0 || i >= n) {
            PyErr_SetString(PyExc_IndexError, "copy_pairs: index out of range");
            plain = -1;
            break;
        }
        PyObject* pair = PyList_GET_ITEM(pairs, i);
        if (!PyTuple_CheckExact(pair) || PyTuple_GET_SIZE(pair) != 2 || (PyObject*)Py_TYPE(PyTuple_GET_ITEM(pair, 0)) != cls ||
            (PyObject*)Py_TYPE(PyTuple_GET_ITEM(pair, 1)) != cls) {
            plain = 0;
            break;
        }
        PyObject* a = copy_instance(type, PyTuple_GET_ITEM(pair, 0), empty);
        PyObject* b = a ? copy_instance(type, PyTuple_GET_ITEM(pair, 1), empty) : NULL;
        PyObject* copy = (a && b) ? PyTuple_Pack(2, a, b) : NULL;
        Py_XDECREF(a);
        Py_XDECREF(b);
        if (copy == NULL) {
            plain = PyErr_Occurred() ? -1 : 0;
            break;
        }
        PyList_SET_ITEM(out, k, copy);
    }
    Py_DECREF(empty);
    PyBuffer_Release(&view);
    if (plain == 1) return out;
    Py_DECREF(out);
    if (plain < 0) return NULL;
    Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"match_pairs", match_pairs, METH_VARARGS, "[(features_a[m.a_index], features_b[m.b_index]) for m in matches], or None"},
    {"pair_arrays", pair_arrays, METH_VARARGS, "fill out[2][n][2] with the x, y of (cls, cls) pairs; False if the list is not plain"},
    {"copy_pairs", copy_pairs, METH_VARARGS, "[deepcopy(pairs[i]) for i in order] for plain (cls, cls) pairs, or None"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_sfm_hostfast", "bulk conversions of the drop-in boundary", -1, methods};

PyMODINIT_FUNC PyInit__sfm_hostfast(void) {
    s_x = PyUnicode_InternFromString("x");
    s_y = PyUnicode_InternFromString("y");
    s_a_index = PyUnicode_InternFromString("a_index");
    s_b_index = PyUnicode_InternFromString("b_index");
    if (!s_x || !s_y || !s_a_index || !s_b_index) return NULL;
    return PyModule_Create(&module);
}
