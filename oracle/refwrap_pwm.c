/*
 * refwrap_pwm.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Builds the reference's own window-scoring loop (the static `calculate` of
 * rnascan/BioAddons/motifs/_pwm.c:7-70) from the source file WHERE IT LIES
 * under /root/reference: the file is pulled in with #include (path given by
 * -DREF_PWM_C=...), nothing from it is copied into this repository and no
 * header or library is stood in for (Python.h and the numpy headers are part
 * of this image).
 *
 * Why a wrapper at all: the reference's Python entry point py_calculate
 * (_pwm.c:79-121) parses "s#" into an `int` (_pwm.c:86), which CPython >= 3.10
 * rejects (PY_SSIZE_T_CLEAN) -- every call raises SystemError.  The arithmetic
 * we want to pin is in the static function, so this file registers its own
 * module `_refpwm` whose calculate() parses arguments correctly, applies the
 * same dtype/rank/column checks as _pwm.c:96-113 and then calls the
 * reference's `calculate` unchanged.
 *
 * Output goes to oracle/_ref/ only (git-ignored, travels with gpurun).
 */
#define PY_SSIZE_T_CLEAN
#include REF_PWM_C   /* defines static calculate(), py_calculate(), PyInit__pwm */

static PyObject *
refwrap_calculate(PyObject *self, PyObject *args)
{
    const char *sequence;
    Py_ssize_t s;
    PyObject *matrix = NULL;
    PyObject *result;
    PyArrayObject *array;
    if (!PyArg_ParseTuple(args, "s#O&", &sequence, &s, PyArray_Converter, &matrix))
        return NULL;
    array = (PyArrayObject *)matrix;
    if (PyArray_TYPE(array) != NPY_DOUBLE || PyArray_NDIM(array) != 2 ||
        PyArray_DIM(array, 1) != 4) {
        PyErr_SetString(PyExc_ValueError, "matrix must be float64 [m][4]");
        result = NULL;
    } else if (s > 0x7fffffff) {
        PyErr_SetString(PyExc_ValueError, "sequence too long for the reference's int length");
        result = NULL;
    } else {
        /* the reference's function, untouched: _pwm.c:7-70 */
        result = calculate(sequence, (int)s, matrix, PyArray_DIM(array, 0));
    }
    Py_DECREF(matrix);
    return result;
}

static struct PyMethodDef refwrap_methods[] = {
    {"calculate", (PyCFunction)refwrap_calculate, METH_VARARGS,
     "calculate(sequence, matrix[m][4] float64) -> float32[n] via the reference's _pwm.c loop"},
    {NULL, NULL, 0, NULL}
};

static struct PyModuleDef refwrap_moduledef = {
    PyModuleDef_HEAD_INIT, "_refpwm",
    "reference _pwm.c calculate(), compiled from /root/reference where it lies",
    -1, refwrap_methods, NULL, NULL, NULL, NULL
};

PyMODINIT_FUNC
PyInit__refpwm(void)
{
    import_array();
    return PyModule_Create(&refwrap_moduledef);
}
