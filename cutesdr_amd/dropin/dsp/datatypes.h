// dsp/datatypes.h drop-in: the reference's sample types (reference dsp/datatypes.h:16-45) without
// its <QtGui/QApplication> include.  TYPEREAL stays double: the class surface is unchanged, the
// fp32 conversion happens inside libcutesdr_mi.
#ifndef DATATYPES_H
#define DATATYPES_H
#include <cstdint>
#include <cmath>
#if defined(QT_CORE_LIB) || defined(QT_VERSION)
#include <QtGlobal>
#else
typedef int16_t qint16;
typedef int32_t qint32;
#endif

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
