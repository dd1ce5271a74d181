"""``name value`` command line parameters, numbered run directories and the stdout tee:
the interface of the reference's ``tools_wscale/paramhelpers.py`` (getParam :16-27,
checkUnusedParams :29-37, writeParams :41-45, getNextGenericPath :65-74, Logger :83-94)."""
import json
import os
import sys

paramUsed = []
paramDict = {}


def getParam(name, default, argv=None):
    """value following `name` on the command line (case-insensitive), else default; strings are
    returned unconverted, as in the reference"""
    argv = sys.argv if argv is None else argv
    while len(paramUsed) < len(argv):
        paramUsed.append(0)
    v = default
    for i in range(1, len(argv)):
        if argv[i].lower() == name.lower() and i + 1 < len(paramUsed):
            paramUsed[i] = paramUsed[i + 1] = 1
            v = argv[i + 1]
    paramDict[name] = v
    return v


def checkUnusedParams(argv=None):
    argv = sys.argv if argv is None else argv
    err = False
    for i in range(1, len(argv)):
        if i >= len(paramUsed) or paramUsed[i] == 0:
            print("Error: param %d '%s' not used!" % (i, argv[i]))
            err = True
    if err:
        exit(1)


def writeParams(filename="params.json", data=None):
    with open(filename, "w") as f:
        json.dump(paramDict if data is None else data, f, indent=4)


def readParams(filename="params.json"):
    with open(filename, "r") as f:
        return json.load(f)


def paramsToString():
    return "".join("\t{}: {}\n".format(k, v) for k, v in paramDict.items())


def getNextGenericPath(dirPrefix, folder_no=1, basePath="../data/"):
    while os.path.exists(basePath + "%s_%04d/" % (dirPrefix, folder_no)):
        folder_no += 1
    test_path = basePath + "%s_%04d/" % (dirPrefix, folder_no)
    print("Using %s dir '%s'" % (dirPrefix, test_path))
    os.makedirs(test_path)
    return (test_path, folder_no)


def getNextTestPath(folder_no=1, basePath="../data/"):
    return getNextGenericPath("test", folder_no, basePath)


def getNextSimPath(folder_no=1, basePath="../data/"):
    return getNextGenericPath("sim", folder_no, basePath)


class Logger(object):
    """tee of stdout into <test_path>/logfile.log"""

    def __init__(self, test_path):
        self.terminal = sys.stdout
        self.log = open(test_path + "logfile.log", "a")

    def write(self, message):
        self.terminal.write(message)
        self.log.write(message)

    def flush(self):
        pass
