// tagitem.hpp -- tag lists: the argument passing convention of class JPEG and of the hooks.
// Same identifiers, layout ({JPG_Tag; union{long,float,ptr}}) and control tags as the reference's
// interface/tagitem.hpp:72-290, so client code written against the reference compiles unchanged.
#ifndef MIJ_INTERFACE_TAGITEM_HPP
#define MIJ_INTERFACE_TAGITEM_HPP
#include "jpgtypes.hpp"

typedef JPG_ULONG JPG_Tag;

// control tags (tagitem.hpp:77-103 of the reference)
#define JPGTAG_TAG_DONE (0L)   // ends a list
#define JPGTAG_TAG_END (0L)
#define JPGTAG_TAG_IGNORE (1L) // skip this item
#define JPGTAG_TAG_MORE (2L)   // continue at the list ti_pPtr points to
#define JPGTAG_TAG_SKIP (3L)   // skip this and the next ti_lData items
#define JPGTAG_TAG_USER (((JPG_ULONG)1) << 31)
#define JPGTAG_SET (((JPG_ULONG)1) << 30)

#define JPG_PointerTag(id, ptr) JPG_TagItem(id, (JPG_APTR)(ptr))
#define JPG_ValueTag(id, v) JPG_TagItem(id, (JPG_LONG)(v))
#define JPG_FloatTag(id, f) JPG_TagItem(id, (JPG_FLOAT)(f))
#define JPG_Continue(tag) JPG_TagItem(JPGTAG_TAG_MORE, const_cast<struct JPG_TagItem *>(tag))
#define JPG_EndTag JPG_TagItem(JPGTAG_TAG_DONE)

struct JPG_EXPORT JPG_TagItem {
  JPG_Tag ti_Tag;
  union TagContents {
    JPG_LONG ti_lData;
    JPG_FLOAT ti_fData;
    JPG_APTR ti_pPtr;
    TagContents(JPG_LONG v) : ti_pPtr(0) { ti_lData = v; }
    TagContents(JPG_FLOAT v) : ti_pPtr(0) { ti_fData = v; }
    TagContents(JPG_APTR p) : ti_pPtr(p) {}
    TagContents() {}
  } ti_Data;

  JPG_TagItem(JPG_Tag tag, JPG_LONG data) : ti_Tag(tag), ti_Data(data) {}
  JPG_TagItem(JPG_Tag tag, JPG_FLOAT data) : ti_Tag(tag), ti_Data(data) {}
  JPG_TagItem(JPG_Tag tag, JPG_APTR ptr = 0) : ti_Tag(tag), ti_Data(ptr) {}
  JPG_TagItem() {}

  // The next real item after this one (control tags resolved), or NULL at the end of the list.
  struct JPG_TagItem *NextTagItem()
  {
    struct JPG_TagItem *t = this + 1;
    return Resolve(t);
  }
  const struct JPG_TagItem *NextTagItem() const { return const_cast<JPG_TagItem *>(this)->NextTagItem(); }

  // This item if it is a real one, else the first real item after it (extension: the reference has no name for it).
  struct JPG_TagItem *FirstTagItem() { return Resolve(this); }
  const struct JPG_TagItem *FirstTagItem() const { return Resolve(const_cast<JPG_TagItem *>(this)); }

  struct JPG_TagItem *FindTagItem(JPG_Tag id)
  {
    for (struct JPG_TagItem *t = Resolve(this); t; t = t->NextTagItem())
      if ((t->ti_Tag & ~JPGTAG_SET) == id) return t;
    return 0;
  }
  const struct JPG_TagItem *FindTagItem(JPG_Tag id) const { return const_cast<JPG_TagItem *>(this)->FindTagItem(id); }

  JPG_LONG GetTagData(JPG_Tag id, JPG_LONG def = 0) const
  {
    const struct JPG_TagItem *t = FindTagItem(id);
    return t ? t->ti_Data.ti_lData : def;
  }
  JPG_FLOAT GetTagFloat(JPG_Tag id, JPG_FLOAT def = 0.0f) const
  {
    const struct JPG_TagItem *t = FindTagItem(id);
    return t ? t->ti_Data.ti_fData : def;
  }
  JPG_APTR GetTagPtr(JPG_Tag id, JPG_APTR def = 0) const
  {
    const struct JPG_TagItem *t = FindTagItem(id);
    return t ? t->ti_Data.ti_pPtr : def;
  }
  void SetTagData(JPG_Tag id, JPG_LONG v)
  {
    if (struct JPG_TagItem *t = FindTagItem(id)) t->ti_Data.ti_lData = v;
  }
  void SetTagFloat(JPG_Tag id, JPG_FLOAT v)
  {
    if (struct JPG_TagItem *t = FindTagItem(id)) t->ti_Data.ti_fData = v;
  }
  void SetTagPtr(JPG_Tag id, JPG_APTR p)
  {
    if (struct JPG_TagItem *t = FindTagItem(id)) t->ti_Data.ti_pPtr = p;
  }

private:
  // follow control tags starting AT t
  static struct JPG_TagItem *Resolve(struct JPG_TagItem *t)
  {
    while (t) {
      switch (t->ti_Tag) {
      case JPGTAG_TAG_DONE: return 0;
      case JPGTAG_TAG_IGNORE: t++; break;
      case JPGTAG_TAG_MORE: t = (struct JPG_TagItem *)t->ti_Data.ti_pPtr; break;
      case JPGTAG_TAG_SKIP: t += 1 + t->ti_Data.ti_lData; break;
      default: return t;
      }
    }
    return 0;
  }
};
#endif
