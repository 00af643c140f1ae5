"""Usage banners of the `vapor` command (vapor_vali/prep.pyx in the reference)."""
VERSION = "vapor (MI355X/HIP build) - interface of VaPoR V1.0"

_COMMON = [
    "	--output-path:		folder where the recurrence plots will be kept",
    "	--reference:		reference genome that the PacBio files are aligned against",
    "	--pacbio-input:		absolute path of the input PacBio BAM (XXX or * in the name = one file per chromosome)",
]


def _show(usage, first):
    print(VERSION)
    print("")
    print("Usage: " + usage)
    print("Parameters:")
    for line in [first] + _COMMON:
        print(line)


def print_read_me():
    print(VERSION)
    print("")
    print("Usage: vapor [Options] [Parameters]")
    print("Options: ")
    for o in ("vcf", "bed", "ins"):
        print("	" + o)
    print("Parameters:")
    for line in ["	--sv-input:		input file in bed or vcf format"] + _COMMON:
        print(line)


def readme_bed():
    _show("vapor bed [Parameters]", "	--sv-input:		BED: chr start end SVID TYPE [inserted sequence]; --output-file: name of the result table")


def readme_vcf():
    _show("vapor vcf [Parameters]", "	--sv-input:		input file in vcf format (result: <input>.vapor)")


def readme_melt():
    _show("vapor ins [Parameters]", "	--sv-input:		prefix of the MELT <prefix>.vcf / <prefix>.fa pair")
