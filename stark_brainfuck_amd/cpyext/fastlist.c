/* fastlist.c -- CPython helper for the list-of-objects call surface of the reference.
 *
 * The reference's functions take and return Python lists of element objects (ntt.py:4-42 `values`, fri.py:26-37 `codeword`,
 * merkle.py:8 `data_array`).  Walking such a list in Python costs ~1 us per element -- 4 ms for 2^22 elements each way, more than
 * the transform itself -- so the two conversions live here:
 *     pack_base(seq, out)                         out[i]        = seq[i].value                      (BaseFieldElement)
 *     pack_ext(seq, out)                          out[k*n + i]  = seq[i].polynomial.coefficients[k].value, missing -> 0
 *     unpack_base(buf, cls, field)                [cls-instance(value = buf[i], field = field) ...]  without running __init__
 *     unpack_ext(buf, xcls, pcls, bcls, xfield, bfield)   extension elements with trimmed coefficient lists, as
 *                                                 ExtensionFieldElement.__init__ leaves them (extension_field.py:5-9)
 * `out` / `buf` are C-contiguous uint64 buffers (numpy arrays).  Values outside [0, 2^64) raise OverflowError and the caller
 * falls back to the Python loop (which reduces mod p).  Host-side plumbing only: nothing here touches the GPU.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *s_value, *s_field, *s_polynomial, *s_coefficients;

/* attribute `name` of obj: from the instance dictionary when there is one (elements are plain objects with a __dict__: a
 * dictionary look-up is a third of the generic attribute protocol), else through PyObject_GetAttr.  New reference. */
static PyObject* attr_of(PyObject* obj, PyObject* name) {
    PyObject** dictptr = _PyObject_GetDictPtr(obj);
    if (dictptr && *dictptr) {
        PyObject* v = PyDict_GetItemWithError(*dictptr, name);
        if (v) { Py_INCREF(v); return v; }
        if (PyErr_Occurred()) return NULL;
    }
    return PyObject_GetAttr(obj, name);
}

static int get_u64(PyObject* obj, uint64_t* out) {
    PyObject* v = attr_of(obj, s_value);
    if (!v) return -1;
    unsigned long long x = PyLong_AsUnsignedLongLong(v);
    Py_DECREF(v);
    if (x == (unsigned long long)-1 && PyErr_Occurred()) return -1;
    *out = (uint64_t)x;
    return 0;
}

static int get_out(PyObject* arg, Py_buffer* view, Py_ssize_t need_words, int writable) {
    if (PyObject_GetBuffer(arg, view, writable ? (PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) : PyBUF_C_CONTIGUOUS) < 0) return -1;
    if (view->len < need_words * 8) {
        PyBuffer_Release(view);
        PyErr_SetString(PyExc_ValueError, "buffer too small");
        return -1;
    }
    return 0;
}

static PyObject* pack_base(PyObject* self, PyObject* args) {
    PyObject *seq_in, *out;
    if (!PyArg_ParseTuple(args, "OO", &seq_in, &out)) return NULL;
    PyObject* seq = PySequence_Fast(seq_in, "expected a sequence of elements");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    Py_buffer view;
    if (get_out(out, &view, n, 1) < 0) { Py_DECREF(seq); return NULL; }
    uint64_t* dst = (uint64_t*)view.buf;
    PyObject** items = PySequence_Fast_ITEMS(seq);
    for (Py_ssize_t i = 0; i < n; ++i)
        if (get_u64(items[i], &dst[i]) < 0) { PyBuffer_Release(&view); Py_DECREF(seq); return NULL; }
    PyBuffer_Release(&view);
    Py_DECREF(seq);
    return PyLong_FromSsize_t(n);
}

static PyObject* pack_ext(PyObject* self, PyObject* args) {
    PyObject *seq_in, *out;
    if (!PyArg_ParseTuple(args, "OO", &seq_in, &out)) return NULL;
    PyObject* seq = PySequence_Fast(seq_in, "expected a sequence of elements");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    Py_buffer view;
    if (get_out(out, &view, 3 * n, 1) < 0) { Py_DECREF(seq); return NULL; }
    uint64_t* dst = (uint64_t*)view.buf;
    PyObject** items = PySequence_Fast_ITEMS(seq);
    int failed = 0;
    for (Py_ssize_t i = 0; i < n && !failed; ++i) {
        dst[i] = dst[n + i] = dst[2 * n + i] = 0;
        PyObject* poly = attr_of(items[i], s_polynomial);
        if (!poly) { failed = 1; break; }
        PyObject* coeffs = attr_of(poly, s_coefficients);
        Py_DECREF(poly);
        if (!coeffs) { failed = 1; break; }
        PyObject* cf = PySequence_Fast(coeffs, "coefficients must be a sequence");
        Py_DECREF(coeffs);
        if (!cf) { failed = 1; break; }
        const Py_ssize_t k = PySequence_Fast_GET_SIZE(cf);
        if (k > 3) { PyErr_SetString(PyExc_ValueError, "extension element with more than three coefficients"); failed = 1; }
        for (Py_ssize_t c = 0; c < k && !failed; ++c)
            if (get_u64(PySequence_Fast_GET_ITEM(cf, c), &dst[c * n + i]) < 0) failed = 1;
        Py_DECREF(cf);
    }
    PyBuffer_Release(&view);
    Py_DECREF(seq);
    if (failed) return NULL;
    return PyLong_FromSsize_t(n);
}

/* an instance of `cls` with exactly the attributes value and field, without calling __init__ */
static PyObject* new_base(PyTypeObject* cls, PyObject* empty, uint64_t value, PyObject* field) {
    PyObject* obj = cls->tp_new(cls, empty, NULL);
    if (!obj) return NULL;
    PyObject* v = PyLong_FromUnsignedLongLong(value);
    if (!v || PyObject_SetAttr(obj, s_value, v) < 0 || PyObject_SetAttr(obj, s_field, field) < 0) { Py_XDECREF(v); Py_DECREF(obj); return NULL; }
    Py_DECREF(v);
    return obj;
}

static PyObject* unpack_base(PyObject* self, PyObject* args) {
    PyObject *buf, *cls, *field;
    if (!PyArg_ParseTuple(args, "OOO", &buf, &cls, &field)) return NULL;
    if (!PyType_Check(cls)) { PyErr_SetString(PyExc_TypeError, "cls must be a class"); return NULL; }
    Py_buffer view;
    if (get_out(buf, &view, 0, 0) < 0) return NULL;
    const Py_ssize_t n = view.len / 8;
    const uint64_t* src = (const uint64_t*)view.buf;
    PyObject* empty = PyTuple_New(0);
    PyObject* list = PyList_New(n);
    /* a million new container objects would run the cyclic collector hundreds of times, each pass walking what has been built
     * so far (1.3 us per element measured, against 0.25 us with the collector paused): nothing built here can be garbage */
    const int gc_was_on = PyGC_Disable();
    for (Py_ssize_t i = 0; list && i < n; ++i) {
        PyObject* obj = new_base((PyTypeObject*)cls, empty, src[i], field);
        if (!obj) { Py_CLEAR(list); break; }
        PyList_SET_ITEM(list, i, obj);
    }
    if (gc_was_on) PyGC_Enable();
    Py_XDECREF(empty);
    PyBuffer_Release(&view);
    return list;
}

static PyObject* unpack_ext(PyObject* self, PyObject* args) {
    PyObject *buf, *xcls, *pcls, *bcls, *xfield, *bfield;
    if (!PyArg_ParseTuple(args, "OOOOOO", &buf, &xcls, &pcls, &bcls, &xfield, &bfield)) return NULL;
    if (!PyType_Check(xcls) || !PyType_Check(pcls) || !PyType_Check(bcls)) { PyErr_SetString(PyExc_TypeError, "classes expected"); return NULL; }
    Py_buffer view;
    if (get_out(buf, &view, 0, 0) < 0) return NULL;
    const Py_ssize_t n = view.len / 24;
    const uint64_t* src = (const uint64_t*)view.buf;
    PyObject* empty = PyTuple_New(0);
    PyObject* list = PyList_New(n);
    const int gc_was_on = PyGC_Disable();
    for (Py_ssize_t i = 0; list && i < n; ++i) {
        const uint64_t c[3] = {src[i], src[n + i], src[2 * n + i]};
        const Py_ssize_t k = c[2] ? 3 : (c[1] ? 2 : (c[0] ? 1 : 0));      /* stored coefficients: up to the leading non-zero one */
        PyObject* coeffs = PyList_New(k);
        PyObject* poly = NULL;
        PyObject* x = NULL;
        int ok = coeffs != NULL;
        for (Py_ssize_t j = 0; ok && j < k; ++j) {
            PyObject* b = new_base((PyTypeObject*)bcls, empty, c[j], bfield);
            if (!b) ok = 0; else PyList_SET_ITEM(coeffs, j, b);
        }
        if (ok) { poly = ((PyTypeObject*)pcls)->tp_new((PyTypeObject*)pcls, empty, NULL); ok = poly && PyObject_SetAttr(poly, s_coefficients, coeffs) == 0; }
        if (ok) { x = ((PyTypeObject*)xcls)->tp_new((PyTypeObject*)xcls, empty, NULL); ok = x && PyObject_SetAttr(x, s_polynomial, poly) == 0 && PyObject_SetAttr(x, s_field, xfield) == 0; }
        Py_XDECREF(coeffs);
        Py_XDECREF(poly);
        if (!ok) { Py_XDECREF(x); Py_CLEAR(list); break; }
        PyList_SET_ITEM(list, i, x);
    }
    if (gc_was_on) PyGC_Enable();
    Py_XDECREF(empty);
    PyBuffer_Release(&view);
    return list;
}

static PyMethodDef methods[] = {
    {"pack_base", pack_base, METH_VARARGS, "values of a sequence of base-field elements -> uint64 buffer"},
    {"pack_ext", pack_ext, METH_VARARGS, "limbs of a sequence of extension-field elements -> uint64 buffer of 3 planes"},
    {"unpack_base", unpack_base, METH_VARARGS, "uint64 buffer -> list of base-field element objects"},
    {"unpack_ext", unpack_ext, METH_VARARGS, "uint64 buffer of 3 planes -> list of extension-field element objects"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fastlist", "list <-> uint64 buffer conversions for element objects", -1, methods};

PyMODINIT_FUNC PyInit__fastlist(void) {
    s_value = PyUnicode_InternFromString("value");
    s_field = PyUnicode_InternFromString("field");
    s_polynomial = PyUnicode_InternFromString("polynomial");
    s_coefficients = PyUnicode_InternFromString("coefficients");
    return PyModule_Create(&moduledef);
}
