// tests/qtshim/qtshim_core.h -- TEST INFRASTRUCTURE: a minimal stand-in for the Qt 6 Core classes that uvgComm's KvazaarFilter / OpenHEVCFilter
// translation units touch (QString, QSettings, QSize, QThread, QMutex, ...), just enough for
//     g++ -std=c++17 -fsyntax-only -I include -I tests/qtshim -I /root/reference/src ... /root/reference/src/media/processing/kvazaarfilter.cpp
// to type-check the reference's OWN source files, where they lie, against THIS repository's include/kvazaar.h and include/openHevcWrapper.h
// (tests/test_reference_compiles.py; SURVEY.md section 8(b): the drop-in boundary).  Nothing here is linked or run, nothing of the reference
// is copied: declarations only, with the signatures Qt documents.  Not part of the product.
#pragma once
#include <cstdint>
#include <cstring>      // (Qt's own headers bring <cstring> in: the reference's files call memcpy without including it)
#include <deque>
#include <initializer_list>
#include <memory>
#include <string>
#include <vector>

typedef unsigned char uchar;
typedef unsigned short quint16;
typedef unsigned int quint32;
typedef long long qint64;
typedef unsigned long long quint64;

#define Q_OBJECT
#define Q_ASSERT(c) ((void)(c))
#define Q_UNUSED(x) (void)x;
#define signals public
#define slots
#define emit

class QByteArray {
 public:
  QByteArray();
  QByteArray(const char *);
  operator const char *() const;          // (Qt: available unless QT_NO_CAST_FROM_BYTEARRAY; uvgComm relies on it, kvazaarfilter.cpp:172-174)
  const char *data() const;
  const char *constData() const;
  int size() const;
};

class QString {
 public:
  QString();
  QString(const char *);
  QString(const QByteArray &);
  QString(const std::string &) = delete;
  static QString number(int, int base = 10);
  static QString number(unsigned int, int base = 10);
  static QString number(long, int base = 10);
  static QString number(unsigned long, int base = 10);
  static QString number(long long, int base = 10);
  static QString number(unsigned long long, int base = 10);
  static QString number(double, char format = 'g', int precision = 6);
  static QString fromStdString(const std::string &);
  QByteArray toUtf8() const;
  QByteArray toLocal8Bit() const;
  QByteArray toLatin1() const;
  std::string toStdString() const;
  int toInt(bool *ok = nullptr, int base = 10) const;
  unsigned int toUInt(bool *ok = nullptr, int base = 10) const;
  double toDouble(bool *ok = nullptr) const;
  bool isEmpty() const;
  int size() const;
  int length() const;
  QString &operator+=(const QString &);
  QString &operator+=(const char *);
  QString arg(const QString &) const;
  QString arg(int) const;
};
QString operator+(const QString &, const QString &);
QString operator+(const QString &, const char *);
QString operator+(const char *, const QString &);
bool operator==(const QString &, const QString &);
bool operator==(const QString &, const char *);
bool operator==(const char *, const QString &);
bool operator!=(const QString &, const QString &);
bool operator!=(const QString &, const char *);
bool operator<(const QString &, const QString &);

template <class T> class QList {
 public:
  QList();
  QList(std::initializer_list<T>);
  int size() const;
  bool isEmpty() const;
  const T &at(int) const;
  T &operator[](int);
  const T &operator[](int) const;
  void append(const T &);
  void push_back(const T &);
  QList &operator<<(const T &);
  const T *begin() const;
  const T *end() const;
};
class QStringList : public QList<QString> {
 public:
  QStringList();
  QStringList(std::initializer_list<QString>);
};

class QVariant {
 public:
  QVariant();
  QVariant(int);
  QVariant(const QString &);
  QVariant(const char *);
  int toInt(bool *ok = nullptr) const;
  unsigned int toUInt(bool *ok = nullptr) const;
  double toDouble(bool *ok = nullptr) const;
  bool toBool() const;
  QString toString() const;
  bool isValid() const;
  bool isNull() const;
};
bool operator==(const QVariant &, const QVariant &);
bool operator==(const QVariant &, const char *);
bool operator==(const QVariant &, const QString &);
bool operator!=(const QVariant &, const QVariant &);
bool operator!=(const QVariant &, const char *);
bool operator!=(const QVariant &, const QString &);

class QSettings {
 public:
  enum Format { NativeFormat, IniFormat };
  QSettings(const QString &fileName, Format format);
  QVariant value(const QString &key) const;
  QVariant value(const QString &key, const QVariant &defaultValue) const;
  void setValue(const QString &key, const QVariant &value);
  int beginReadArray(const QString &prefix);
  void setArrayIndex(int i);
  void endArray();
  bool contains(const QString &key) const;
  void sync();
};

class QSize {
 public:
  QSize();
  QSize(int w, int h);
  int width() const;
  int height() const;
};

class QObject {
 public:
  explicit QObject(QObject *parent = nullptr);
  virtual ~QObject();
};

class QMutex {
 public:
  QMutex();
  ~QMutex();
  void lock();
  bool tryLock(int timeout = 0);
  void unlock();
};

class QWaitCondition {
 public:
  QWaitCondition();
  ~QWaitCondition();
  bool wait(QMutex *lockedMutex, unsigned long time = ~0ul);
  void wakeOne();
  void wakeAll();
};

class QThread : public QObject {
 public:
  explicit QThread(QObject *parent = nullptr);
  ~QThread() override;
  void start();
  void quit();
  bool wait(unsigned long time = ~0ul);
  bool isRunning() const;
  static int idealThreadCount();
  static void sleep(unsigned long);
  static void msleep(unsigned long);
  static void usleep(unsigned long);
  static QThread *currentThread();
 protected:
  virtual void run();
};

class QFile {
 public:
  QFile();
  explicit QFile(const QString &name);
};
class QTextStream {
 public:
  QTextStream();
};

class QDateTime {
 public:
  static qint64 currentMSecsSinceEpoch();
};
class QTime {
 public:
  static QTime currentTime();
};

class QHostAddress {
 public:
  QHostAddress();
  explicit QHostAddress(const QString &);
  QString toString() const;
};

class QDebug {
 public:
  QDebug &noSpace();
  template <class T> QDebug &operator<<(const T &);
};
QDebug qDebug();
QDebug qCritical();
QDebug qWarning();
