// dsp/datatypes.h drop-in: the reference's sample types (reference dsp/datatypes.h:11-45).  Like the
// reference it pulls in Qt's application header (dsp/datatypes.h:11 -- host code gets qint16/qint32,
// QString ... through it) whenever Qt is on the include path; without Qt (plain g++ hosts, the tests) the two
// integer typedefs the class surface needs come from <cstdint>.  TYPEREAL stays double: the class surface is
// unchanged, the fp32 conversion happens inside libcutesdr_mi.
#ifndef DATATYPES_H
#define DATATYPES_H
#if defined(__has_include)
#if __has_include(<QtGui/QApplication>)
#include <QtGui/QApplication>
#define CSDR_DROPIN_QT 1
#elif __has_include(<QtWidgets/QApplication>)
#include <QtWidgets/QApplication>
#define CSDR_DROPIN_QT 1
#elif __has_include(<QtGlobal>)
#include <QtGlobal>
#define CSDR_DROPIN_QT 1
#endif
#endif
#if !defined(CSDR_DROPIN_QT) && (defined(QT_CORE_LIB) || defined(QT_VERSION))
#include <QtGlobal>
#define CSDR_DROPIN_QT 1
#endif
#ifndef CSDR_DROPIN_QT
#include <cstdint>
typedef int16_t qint16;
typedef int32_t qint32;
#endif
#include <math.h>

typedef float tSReal;
typedef double tDReal;
typedef struct _sCplx { tSReal re; tSReal im; } tSComplex;
typedef struct _dCplx { tDReal re; tDReal im; } tDComplex;
typedef struct _isCplx { qint16 re; qint16 im; } tStereo16;

#define TYPEREAL tDReal
#define TYPECPX tDComplex
#define TYPESTEREO16 tStereo16
#define TYPEMONO16 qint16
#define K_2PI (2.0 * 3.14159265358979323846)
#define K_PI (3.14159265358979323846)

#ifndef TRUE
#define TRUE true
#define FALSE false
#endif
#endif  // DATATYPES_H
