# Generates EbApiVersion.h from the reference's own template with cmake's configure_file -- the same
# step the reference performs at CMakeLists.txt:53 (version numbers from CMakeLists.txt:49-51).
# Output goes to oracle/_ref/gen/ (git-ignored).  Nothing is written under /root/reference.
set(SVT_VP9_VERSION_MAJOR "0")
set(SVT_VP9_VERSION_MINOR "3")
set(SVT_VP9_VERSION_PATCHLEVEL "1")
configure_file(${REF}/Source/API/EbApiVersion.h.in ${OUT}/EbApiVersion.h @ONLY)
